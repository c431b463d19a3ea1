// TEST INFRASTRUCTURE ONLY -- never linked into or imported by the product path.
//
// Host-side shim that lets the reference's CuPy ElementwiseKernel bodies (CUDA-C strings living in
// /root/reference/.../kernels/*.py) be compiled *unmodified* with g++ and executed sequentially.
// oracle/build_ref.py pulls the strings out of the reference at build time, wraps each one in
//     extern "C" void <name>(<raw arrays...>, long size) { for (ptrdiff_t i = 0; i < size; ++i) { <operation> } }
// and writes the generated translation unit + .so under oracle/_ref/ (git-ignored, never committed).
//
// What is modelled here is the part of CuPy's kernel prelude (cupy/_core/include/cupy/carray.cuh,
// not vendored in the reference) that these kernels rely on:
//   * class float16: storage = IEEE binary16, implicit ctor from float (round-to-nearest-even),
//     explicit ctors from double/int (via float), implicit `operator float()`, templated compound
//     assignment `x += r  ==>  x = x + r`, and min/max overloads returning float16;
//   * atomicAdd on float*/unsigned*/int* (sequential execution => plain read-modify-write);
//   * CArray raw indexing `a[expr]` where expr may be a float (truncated to ptrdiff_t);
//   * CUDA math overloads min/max(float,double), __float_as_uint, __uint_as_float.
// Residual uncertainty (CuPy's header is not available offline) is listed in DESIGN.md "Oracle".
#pragma once
#include <immintrin.h>
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <algorithm>

#define __device__
#define __forceinline__ inline

using std::abs;
using std::fabs;
using std::sqrt;
using std::min;
using std::max;

struct float16 {
  uint16_t d;
  float16() {}
  float16(float v) : d(_cvtss_sh(v, _MM_FROUND_TO_NEAREST_INT | _MM_FROUND_NO_EXC)) {}
  explicit float16(double v) : d(_cvtss_sh((float)v, _MM_FROUND_TO_NEAREST_INT | _MM_FROUND_NO_EXC)) {}
  explicit float16(int v) : d(_cvtss_sh((float)v, _MM_FROUND_TO_NEAREST_INT | _MM_FROUND_NO_EXC)) {}
  explicit float16(bool v) : d(_cvtss_sh((float)v, _MM_FROUND_TO_NEAREST_INT | _MM_FROUND_NO_EXC)) {}
  operator float() const { return _cvtsh_ss(d); }
  template <class X> float16& operator+=(const X& r) { *this = *this + r; return *this; }
  template <class X> float16& operator-=(const X& r) { *this = *this - r; return *this; }
  template <class X> float16& operator*=(const X& r) { *this = *this * r; return *this; }
  template <class X> float16& operator/=(const X& r) { *this = *this / r; return *this; }
};
inline float16 min(float16 x, float16 y) { return float16(std::min(float(x), float(y))); }
inline float16 max(float16 x, float16 y) { return float16(std::max(float(x), float(y))); }

// CUDA's mixed-precision min/max overloads (math_functions.hpp): promote to double, fmin/fmax.
inline double min(float a, double b) { return std::fmin((double)a, b); }
inline double min(double a, float b) { return std::fmin(a, (double)b); }
inline double max(float a, double b) { return std::fmax((double)a, b); }
inline double max(double a, float b) { return std::fmax(a, (double)b); }

inline float atomicAdd(float* a, float v) { float o = *a; *a = o + v; return o; }
inline unsigned atomicAdd(unsigned* a, unsigned v) { unsigned o = *a; *a = o + v; return o; }
inline int atomicAdd(int* a, int v) { int o = *a; *a = o + v; return o; }

inline unsigned __float_as_uint(float f) { unsigned u; std::memcpy(&u, &f, 4); return u; }
inline float __uint_as_float(unsigned u) { float f; std::memcpy(&f, &u, 4); return f; }

template <class X> struct Raw {
  X* p;
  template <class I> X& operator[](I i) const { return p[(ptrdiff_t)i]; }
};
