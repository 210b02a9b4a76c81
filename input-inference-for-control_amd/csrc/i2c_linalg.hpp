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

// ---- scalar math, overloaded on precision ----------------------------------------------------
// The fp64 versions avoid the IEEE divide / sqrt expansions (v_div_scale/fmas/fixup sequences)
// and OCML's branchy large-argument trig path; results are within ~1-2 ulp, far inside the
// parity tolerance. The host simulation runs the SAME formulas; only the hardware seed
// instructions (v_rsq_f64 / v_rcp_f64, ~2^-23 accurate) are emulated by a float-rounded value.
#ifdef I2C_HOST_SIM
I2C_FN double seed_rsq(double x) { return (double)(float)(1.0 / std::sqrt(x)); }
I2C_FN double seed_rcp(double x) { return (double)(float)(1.0 / x); }
I2C_FN double m_fma(double a, double b, double c) { return std::fma(a, b, c); }
I2C_FN double m_rint(double x) { return std::rint(x); }
I2C_FN double m_fabs(double x) { return std::fabs(x); }
I2C_FN float r_rsqrt(float x) { return 1.0f / std::sqrt(x); }
I2C_FN float r_rcp(float x) { return 1.0f / x; }
I2C_FN double r_exp(double x) { return std::exp(x); }
I2C_FN float r_exp(float x) { return std::exp(x); }
I2C_FN double r_log(double x) { return std::log(x); }
I2C_FN float r_log(float x) { return std::log(x); }
I2C_FN void r_sincos(float x, float* s, float* c) { *s = std::sin(x); *c = std::cos(x); }
#else
I2C_FN double seed_rsq(double x) { return __builtin_amdgcn_rsq(x); }
I2C_FN double seed_rcp(double x) { return __builtin_amdgcn_rcp(x); }
I2C_FN double m_fma(double a, double b, double c) { return fma(a, b, c); }
I2C_FN double m_rint(double x) { return rint(x); }
I2C_FN double m_fabs(double x) { return fabs(x); }
I2C_FN float r_rsqrt(float x) { return rsqrtf(x); }
I2C_FN float r_rcp(float x) { return 1.0f / x; }
I2C_FN float r_exp(float x) { return expf(x); }
I2C_FN double r_log(double x) { return log(x); }
I2C_FN float r_log(float x) { return logf(x); }
I2C_FN void r_sincos(float x, float* s, float* c) { sincosf(x, s, c); }
#endif

// Horner step p * z + c with a CONSTANT addend. hipcc otherwise emits v_mov_b64 (copy the constant) +
// v_fmac_f64 (2-address form) for every step; the 3-address v_fma_f64 needs no copy.
#ifdef I2C_HOST_SIM
I2C_FN double p_fma(double p, double z, double c) { return std::fma(p, z, c); }
#else
I2C_FN double p_fma(double p, double z, double c) {
  double r;
  asm("v_fma_f64 %0, %1, %2, %3" : "=v"(r) : "v"(p), "v"(z), "v"(c));
  return r;
}
#endif

// 1/sqrt(x): ~2^-23 seed + one cubically convergent correction (relative error ~ e^3).
I2C_FN double r_rsqrt(double x) {
  const double y = seed_rsq(x);
  const double e = m_fma(-(x * y), y, 1.0);  // 1 - x y^2
  return m_fma(y * e, m_fma(e, 0.375, 0.5), y);
}
// 1/x: seed + two Newton steps.
I2C_FN double r_rcp(double x) {
  double y = seed_rcp(x);
  y = m_fma(m_fma(-x, y, 1.0), y, y);
  return m_fma(m_fma(-x, y, 1.0), y, y);
}
// exp(x) for the pdf ratio / soft policy weight (x <= 0 there): n = rint(x log2 e), r = x - n ln 2 (two-constant Cody-Waite),
// exp(r) = E(r^2) + r O(r^2) with the even / odd halves of the degree-13 Taylor polynomial on |r| <= ln2 / 2 (truncation 4e-18;
// two independent Horner chains: back-to-back dependent inline-asm FMAs get an s_nop each from the hazard recogniser), 2^n by
// v_ldexp_f64 (which also underflows to 0 for very negative x); <= 1.3 ulp (checked against 50-digit arithmetic on 45 000
// arguments in [-745, 0]). 20 instructions against the 31 of the library routine (which copies every coefficient for a
// 2-address v_fmac and guards over/underflow with compare-select chains); NaN in, NaN out.
#ifndef I2C_HOST_SIM
I2C_FN double r_exp(double x) {
  const double n = m_rint(x * 1.44269504088896338700e+00);
  double r = m_fma(-n, 6.93147180369123816490e-01, x);
  r = m_fma(-n, 1.90821492927058770002e-10, r);
  const double z = r * r;
  double e = p_fma(2.08767569878681e-09, z, 2.755731922398589e-07);   // 1/12!, 1/10!
  double o = p_fma(1.6059043836821613e-10, z, 2.505210838544172e-08);  // 1/13!, 1/11!
  e = p_fma(e, z, 2.48015873015873e-05);
  o = p_fma(o, z, 2.7557319223985893e-06);
  e = p_fma(e, z, 1.388888888888889e-03);
  o = p_fma(o, z, 1.984126984126984e-04);
  e = p_fma(e, z, 4.1666666666666664e-02);
  o = p_fma(o, z, 8.333333333333333e-03);
  e = p_fma(e, z, 0.5);
  o = p_fma(o, z, 1.6666666666666666e-01);
  e = m_fma(e, z, 1.0);
  o = m_fma(o, z, 1.0);
  return __builtin_ldexp(m_fma(r, o, e), (int)n);
}
#endif

// sin and cos together, BRANCH-FREE (a branch per call would cut every cell into small scheduling
// regions): three-term Cody-Waite reduction by pi/2 (each n * chunk product is exact for |n| < 2^20,
// i.e. |x| < 1.6e6) + the classic degree-13/14 minimax kernels on [-pi/4, pi/4]; <= 1 ulp there.
// |x| >= 1e6 rad (a diverged trajectory) yields NaN, which the next Cholesky flags in status[b] --
// the reference would still evaluate np.sin exactly there; documented deviation.
I2C_FN void r_sincos(double x, double* s, double* c) {
  // NaN outside the supported range. The NaN must be a CONSTANT: written as (x - x) / (x - x) the compiler guards the
  // IEEE division with a divergent branch (s_and_saveexec / s_cbranch_execz) that splits every cell's scheduling region.
  x = m_fabs(x) < 1.0e6 ? x : __builtin_nan("");
  const double n = m_rint(x * 6.36619772367581382433e-01);
  double r = m_fma(-n, 1.57079632673412561417e+00, x);
  r = m_fma(-n, 6.07710050630396597660e-11, r);
  r = m_fma(-n, 2.02226624871116645580e-21, r);
  const double z = r * r;
  double ps = p_fma(z, 1.58969099521155010221e-10, -2.50507602534068634195e-08);
  ps = p_fma(ps, z, 2.75573137070700676789e-06);
  ps = p_fma(ps, z, -1.98412698298579493134e-04);
  ps = p_fma(ps, z, 8.33333333332248946124e-03);
  const double sr = m_fma(z * r, m_fma(z, ps, -1.66666666666666324348e-01), r);
  double pc = p_fma(z, -1.13596475577881948265e-11, 2.08757232129817482790e-09);
  pc = p_fma(pc, z, -2.75573143513906633035e-07);
  pc = p_fma(pc, z, 2.48015872894767294178e-05);
  pc = p_fma(pc, z, -1.38888888888741095749e-03);
  pc = p_fma(pc, z, 4.16666666666666019037e-02);
  const double cr = 1.0 - m_fma(0.5, z, -(z * z) * pc);
  const int q = (int)n;
  const double ss = (q & 1) ? cr : sr;
  const double cc = (q & 1) ? sr : cr;
  *s = (q & 2) ? -ss : ss;
  *c = ((q + 1) & 2) ? -cc : cc;
}
// The same two routines with every polynomial constant as a SCALAR operand (an s_mov pair next to its use instead of a VGPR
// pair pinned for the whole sweep): for kernels whose registers are full and whose waves share a SIMD, where the scalar unit
// issues beside the vector one (the d = 16 quad forward kernel: the pinned constants were spilled to scratch).
#ifdef I2C_HOST_SIM
I2C_FN void r_sincos_sc(double x, double* s, double* c) { r_sincos(x, s, c); }
I2C_FN double r_exp_sc(double x) { return r_exp(x); }
#else
I2C_FN double p_fma_s(double p, double z, double c) {  // p * z + c, c a scalar operand
  double r;
  asm("v_fma_f64 %0, %1, %2, %3" : "=v"(r) : "v"(p), "v"(z), "s"(c));
  return r;
}
I2C_FN void r_sincos_sc(double x, double* s, double* c) {
  x = m_fabs(x) < 1.0e6 ? x : __builtin_nan("");
  const double n = m_rint(x * 6.36619772367581382433e-01);
  double r = m_fma(-n, 1.57079632673412561417e+00, x);
  r = m_fma(-n, 6.07710050630396597660e-11, r);
  r = m_fma(-n, 2.02226624871116645580e-21, r);
  const double z = r * r;
  double ps = p_fma_s(z * 1.58969099521155010221e-10, 1.0, -2.50507602534068634195e-08);
  ps = p_fma_s(ps, z, 2.75573137070700676789e-06);
  ps = p_fma_s(ps, z, -1.98412698298579493134e-04);
  ps = p_fma_s(ps, z, 8.33333333332248946124e-03);
  const double sr = m_fma(z * r, p_fma_s(z, ps, -1.66666666666666324348e-01), r);
  double pc = p_fma_s(z * -1.13596475577881948265e-11, 1.0, 2.08757232129817482790e-09);
  pc = p_fma_s(pc, z, -2.75573143513906633035e-07);
  pc = p_fma_s(pc, z, 2.48015872894767294178e-05);
  pc = p_fma_s(pc, z, -1.38888888888741095749e-03);
  pc = p_fma_s(pc, z, 4.16666666666666019037e-02);
  const double cr = 1.0 - m_fma(0.5, z, -(z * z) * pc);
  const int q = (int)n;
  const double ss = (q & 1) ? cr : sr;
  const double cc = (q & 1) ? sr : cr;
  *s = (q & 2) ? -ss : ss;
  *c = ((q + 1) & 2) ? -cc : cc;
}
I2C_FN double r_exp_sc(double x) {
  const double n = m_rint(x * 1.44269504088896338700e+00);
  double r = m_fma(-n, 6.93147180369123816490e-01, x);
  r = m_fma(-n, 1.90821492927058770002e-10, r);
  const double z = r * r;
  double e = p_fma_s(z * 2.08767569878681e-09, 1.0, 2.755731922398589e-07);
  double o = p_fma_s(z * 1.6059043836821613e-10, 1.0, 2.505210838544172e-08);
  e = p_fma_s(e, z, 2.48015873015873e-05);
  o = p_fma_s(o, z, 2.7557319223985893e-06);
  e = p_fma_s(e, z, 1.388888888888889e-03);
  o = p_fma_s(o, z, 1.984126984126984e-04);
  e = p_fma_s(e, z, 4.1666666666666664e-02);
  o = p_fma_s(o, z, 8.333333333333333e-03);
  e = p_fma_s(e, z, 0.5);
  o = p_fma_s(o, z, 1.6666666666666666e-01);
  e = m_fma(e, z, 1.0);
  o = m_fma(o, z, 1.0);
  return __builtin_ldexp(m_fma(r, o, e), (int)n);
}
#endif

// ---- polynomial constants held in VGPRs ---------------------------------------------------------------------------
// A VALU instruction reads at most one scalar / literal operand, so a Horner step with a literal coefficient costs the
// lone, issue-bound wave an extra instruction (s_mov pair or v_mov copy) per step. PolyTab keeps the coefficients of
// sincos as opaque per-lane VALUES that the register allocator leaves in VGPRs for the whole sweep (32 registers; only
// worth it where the VGPR file has room, i.e. the small models' forward sweep). Doing the same for exp (15 more
// coefficients) pushed the kernel past 256 VGPRs into scratch (610 us) and was dropped.
template <typename R> struct PolyTab {
  R two_over_pi, pio2_1, pio2_2, pio2_3, s1, s2, s3, s4, s5, s6, c1, c2, c3, c4, c5, c6;
};
template <typename R> I2C_FN R tab_value(double v) {
  R x = (R)v;
#ifndef I2C_HOST_SIM
  asm volatile("" : "+v"(x));
#endif
  return x;
}
template <typename R> I2C_FN void poly_tab_init(PolyTab<R>& t) {
  t.two_over_pi = tab_value<R>(6.36619772367581382433e-01);
  t.pio2_1 = tab_value<R>(1.57079632673412561417e+00);
  t.pio2_2 = tab_value<R>(6.07710050630396597660e-11);
  t.pio2_3 = tab_value<R>(2.02226624871116645580e-21);
  t.s6 = tab_value<R>(1.58969099521155010221e-10);
  t.s5 = tab_value<R>(-2.50507602534068634195e-08);
  t.s4 = tab_value<R>(2.75573137070700676789e-06);
  t.s3 = tab_value<R>(-1.98412698298579493134e-04);
  t.s2 = tab_value<R>(8.33333333332248946124e-03);
  t.s1 = tab_value<R>(-1.66666666666666324348e-01);
  t.c6 = tab_value<R>(-1.13596475577881948265e-11);
  t.c5 = tab_value<R>(2.08757232129817482790e-09);
  t.c4 = tab_value<R>(-2.75573143513906633035e-07);
  t.c3 = tab_value<R>(2.48015872894767294178e-05);
  t.c2 = tab_value<R>(-1.38888888888741095749e-03);
  t.c1 = tab_value<R>(4.16666666666666019037e-02);
}
// r_sincos with the constants taken from the table: the same formulas, instruction for instruction
I2C_FN void r_sincos(double x, const PolyTab<double>& t, double* s, double* c) {
  x = m_fabs(x) < 1.0e6 ? x : __builtin_nan("");
  const double n = m_rint(x * t.two_over_pi);
  double r = m_fma(-n, t.pio2_1, x);
  r = m_fma(-n, t.pio2_2, r);
  r = m_fma(-n, t.pio2_3, r);
  const double z = r * r;
  double ps = p_fma(z, t.s6, t.s5);
  ps = p_fma(ps, z, t.s4);
  ps = p_fma(ps, z, t.s3);
  ps = p_fma(ps, z, t.s2);
  const double sr = m_fma(z * r, p_fma(z, ps, t.s1), r);
  double pc = p_fma(z, t.c6, t.c5);
  pc = p_fma(pc, z, t.c4);
  pc = p_fma(pc, z, t.c3);
  pc = p_fma(pc, z, t.c2);
  pc = p_fma(pc, z, t.c1);
  const double cr = 1.0 - m_fma(0.5, z, -(z * z) * pc);
  const int q = (int)n;
  const double ss = (q & 1) ? cr : sr;
  const double cc = (q & 1) ? sr : cr;
  *s = (q & 2) ? -ss : ss;
  *c = ((q + 1) & 2) ? -cc : cc;
}
I2C_FN void r_sincos(float x, const PolyTab<float>&, float* s, float* c) { r_sincos(x, s, c); }

// the sigma-point offsets d = sf L[i][j] go through the same branch-free routine
I2C_FN void r_sincos_small(double x, double* s, double* c) { r_sincos(x, s, c); }
I2C_FN void r_sincos_small(float x, float* s, float* c) { r_sincos(x, s, c); }

// Global memory access for the [row][B] buffers: a wave-uniform window (buffer resource in SGPRs)
// + a wave-uniform row byte offset (SGPR soffset) + the lane's byte offset (one VGPR, computed
// once per kernel). No per-element 64-bit VALU address arithmetic; rows advance by scalar adds.
#ifdef I2C_HOST_SIM
struct Window {
  char* p;
};
I2C_FN Window make_window(const void* base, unsigned long) { return Window{(char*)base}; }
template <typename R> I2C_FN R wld(const Window& w, unsigned row_off, unsigned lane_off) {
  return *reinterpret_cast<const R*>(w.p + row_off + lane_off);
}
template <typename R> I2C_FN void wst(const Window& w, unsigned row_off, unsigned lane_off, R v) {
  *reinterpret_cast<R*>(w.p + row_off + lane_off) = v;
}
I2C_FN unsigned wld_u8(const Window& w, unsigned off) { return *reinterpret_cast<const unsigned char*>(w.p + off); }
// two consecutive elements in one access (offsets aligned to the pair)
template <typename S> struct Pair2 {
  S a, b;
};
template <typename S> I2C_FN Pair2<S> wld2(const Window& w, unsigned row_off, unsigned lane_off) {
  const S* q = reinterpret_cast<const S*>(w.p + row_off + lane_off);
  return Pair2<S>{q[0], q[1]};
}
template <typename S> I2C_FN void wst2(const Window& w, unsigned row_off, unsigned lane_off, Pair2<S> v) {
  S* q = reinterpret_cast<S*>(w.p + row_off + lane_off);
  q[0] = v.a, q[1] = v.b;
}
#else
struct Window {
  __amdgpu_buffer_rsrc_t r;
};
// `base` and `bytes` must be wave-uniform. A window addresses at most 4 GiB.
// The descriptor inputs go through readfirstlane so that their uniformity is PROVABLE to hipcc: when it
// is not (e.g. after the pointer was spilled through a VGPR), every buffer_load/store gets wrapped in a
// ~10-instruction "waterfall" loop that also serialises consecutive memory operations.
I2C_FN unsigned uniform_u32(unsigned x) { return (unsigned)__builtin_amdgcn_readfirstlane((int)x); }
I2C_FN Window make_window(const void* base, unsigned long bytes) {
  const unsigned long p = (unsigned long)base;
  const unsigned long pu = ((unsigned long)uniform_u32((unsigned)(p >> 32)) << 32) | uniform_u32((unsigned)p);
  const unsigned n = uniform_u32(bytes > 0xFFFFFFFFul ? 0xFFFFFFFFu : (unsigned)bytes);
  return Window{__builtin_amdgcn_make_buffer_rsrc((void*)pu, 0, n, 0x00020000)};
}
template <typename R> I2C_FN R wld(const Window& w, unsigned row_off, unsigned lane_off) {
  if constexpr (sizeof(R) == 8) {
    typedef unsigned v2u __attribute__((ext_vector_type(2)));
    const v2u v = __builtin_amdgcn_raw_buffer_load_b64(w.r, lane_off, uniform_u32(row_off), 0);
    return __builtin_bit_cast(R, v);
  } else {
    return __builtin_bit_cast(R, __builtin_amdgcn_raw_buffer_load_b32(w.r, lane_off, uniform_u32(row_off), 0));
  }
}
// one byte through the buffer path (a global_load_ubyte next to buffer stores makes the waitcnt pass wait for vmcnt(0): it does
// not count on FLAT-encoded and MUBUF operations returning in order with each other)
I2C_FN unsigned wld_u8(const Window& w, unsigned off) { return (unsigned)__builtin_amdgcn_raw_buffer_load_b8(w.r, off, 0, 0); }
I2C_FN void wst(const Window& w, unsigned row_off, unsigned lane_off, double v) {
  typedef unsigned v2u __attribute__((ext_vector_type(2)));
  __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(v2u, v), w.r, lane_off, uniform_u32(row_off), 0);
}
I2C_FN void wst(const Window& w, unsigned row_off, unsigned lane_off, float v) {
  __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), w.r, lane_off, uniform_u32(row_off), 0);
}
// two consecutive elements in one access (doubles: buffer_load / store_dwordx4; floats: dwordx2; offsets aligned to the pair)
template <typename S> struct Pair2 {
  S a, b;
};
template <typename S> I2C_FN Pair2<S> wld2(const Window& w, unsigned row_off, unsigned lane_off) {
  if constexpr (sizeof(S) == 8) {
    typedef unsigned v4u __attribute__((ext_vector_type(4)));
    const v4u v = __builtin_amdgcn_raw_buffer_load_b128(w.r, lane_off, uniform_u32(row_off), 0);
    return __builtin_bit_cast(Pair2<S>, v);
  } else {
    typedef unsigned v2u __attribute__((ext_vector_type(2)));
    const v2u v = __builtin_amdgcn_raw_buffer_load_b64(w.r, lane_off, uniform_u32(row_off), 0);
    return __builtin_bit_cast(Pair2<S>, v);
  }
}
template <typename S> I2C_FN void wst2(const Window& w, unsigned row_off, unsigned lane_off, Pair2<S> v) {
  if constexpr (sizeof(S) == 8) {
    typedef unsigned v4u __attribute__((ext_vector_type(4)));
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4u, v), w.r, lane_off, uniform_u32(row_off), 0);
  } else {
    typedef unsigned v2u __attribute__((ext_vector_type(2)));
    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(v2u, v), w.r, lane_off, uniform_u32(row_off), 0);
  }
}
#endif

// Scheduling fence for the LARGE models only: the cell is thousands of fully unrolled straight-line
// instructions; left alone the scheduler interleaves independent columns / phases for ILP and the
// live ranges overflow 512 VGPRs into scratch. A fence per column / phase bounds the live state.
template <bool ON> I2C_FN void sched_fence() {
#ifndef I2C_HOST_SIM
  if (ON) __builtin_amdgcn_sched_barrier(0);
#endif
}

// Makes a per-lane value opaque to loop-invariant code motion. Products like alpha * sig_xi0[i]
// (alpha fixed per trajectory, sig_xi0 a kernel-argument constant) are otherwise hoisted out of the
// time loop and pinned in dozens of VGPRs for the whole sweep; recomputing them per cell is one FMA.
// Same for a wave-uniform integer (kept in an SGPR): row offsets e * row_bytes are then recomputed by
// one scalar multiply next to each access instead of being hoisted into ~30 pinned SGPRs, which
// otherwise starves the constant operands of the fp64 polynomial kernels of scalar registers.
I2C_FN unsigned opaque_uniform(unsigned x) {
#ifndef I2C_HOST_SIM
  asm volatile("" : "+s"(x));
#endif
  return x;
}
template <typename R> I2C_FN R opaque(R x) {
#ifndef I2C_HOST_SIM
  asm volatile("" : "+v"(x));
#endif
  return x;
}

// clip(x, lo, hi) = min(max(x, lo), hi): two v_max/v_min instead of compare + select chains
#ifdef I2C_HOST_SIM
template <typename R> I2C_FN R r_clip(R x, R lo, R hi) { return x < lo ? lo : (x > hi ? hi : x); }
#else
I2C_FN double r_clip(double x, double lo, double hi) { return fmin(fmax(x, lo), hi); }
I2C_FN float r_clip(float x, float lo, float hi) { return fminf(fmaxf(x, lo), hi); }
#endif

// In-place Cholesky of a packed SPD matrix: a <- L (lower), rinv[j] = 1 / L[j][j].
// Returns false if a pivot is not strictly positive (or NaN): the covariance is not PD.
// Only the LAST pivot is tested: a pivot s_j <= 0 (or NaN) makes 1/sqrt(s_j) NaN or inf, which reaches every later row
// through L[i][j] = v / sqrt(s_j) (0 * inf is NaN too) and turns every later pivot, in particular the last, into NaN or
// -inf -- so `last pivot > 0` is equivalent to `all pivots > 0` at one compare per factorisation instead of one compare
// and one mask update per pivot (the lone wave of the small-batch regime is instruction-issue-bound).
template <int N, typename R> I2C_FN bool chol(R* a, R* rinv) {
  bool ok = true;
#pragma unroll
  for (int j = 0; j < N; ++j) {
    R s = a[tri(j, j)];
#pragma unroll
    for (int k = 0; k < j; ++k) s -= a[tri(j, k)] * a[tri(j, k)];
    if (j == N - 1) ok = s > R(0);
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

// The same for a symmetric NON-SINGULAR matrix that need not be positive definite: a = L Sigma L^T with Sigma = diag(sgn),
// sgn[j] = +-1 the sign of pivot j (unpivoted, so a vanishing leading minor fails like a singular matrix does). fsub / bsub
// apply to the L it returns. Returns false on a zero or NaN pivot.
template <int N, typename R> I2C_FN bool chol_signed(R* a, R* rinv, R* sgn) {
  bool ok = true;
#pragma unroll
  for (int j = 0; j < N; ++j) {
    R s = a[tri(j, j)];
#pragma unroll
    for (int k = 0; k < j; ++k) s -= sgn[k] * a[tri(j, k)] * a[tri(j, k)];
    const R g = s < R(0) ? R(-1) : R(1);
    sgn[j] = g;
    s *= g;
    if (j == N - 1) ok = s > R(0);
    const R r = r_rsqrt(s);
    rinv[j] = r;
    a[tri(j, j)] = s * r;
#pragma unroll
    for (int i = j + 1; i < N; ++i) {
      R v = a[tri(i, j)];
#pragma unroll
      for (int k = 0; k < j; ++k) v -= sgn[k] * a[tri(i, k)] * a[tri(j, k)];
      a[tri(i, j)] = v * r * g;
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
