// sincos_glibc.h — bit-faithful restatement of glibc 2.35 sinf()/cosf() for |x| < 120.
//
// Why: the reference steers the BRIEF pattern with `(float)cos(angle)`, `(float)sin(angle)` on a
// float argument (/root/reference/src/ORBextractor.cc:111-112) which binds to glibc's cosf/sinf.
// Those are NOT correctly rounded (≈0.56 ULP), so a device kernel can only be bit-exact with the
// CPU path by evaluating the very same double-precision polynomial in the very same order.
// Algorithm: glibc sysdeps/ieee754/flt-32/{s_sinf.c,s_cosf.c,s_sincosf.h} (ARM optimized-routines
// sincosf); table constants cross-checked against the bytes of this image's libm.so.6
// (__sincosf_table, see tests/test_sincos.py which also compares against the live libm).
//
// Shared by the HIP kernels (device) and by CPU tests (host). All arithmetic is IEEE double
// without FMA contraction (compile with -ffp-contract=off).
#pragma once
#include <stdint.h>

#ifndef RGBL_HD
#if defined(__HIPCC__)
#define RGBL_HD __host__ __device__ inline
#else
#define RGBL_HD inline
#endif
#endif

namespace rgbl {

RGBL_HD uint32_t sc_asuint(float f) {
  union { float f; uint32_t u; } c; c.f = f; return c.u;
}
RGBL_HD uint32_t sc_abstop12(float x) { return (sc_asuint(x) >> 20) & 0x7ff; }

// n even -> sine polynomial, n odd -> cosine polynomial; `flip` selects the negated cosine table
// (glibc: __sincosf_table[1]).
RGBL_HD float sc_poly(double x, double x2, int n, bool flip) {
  const double c0 = flip ? -0x1p0 : 0x1p0;
  const double c1 = flip ? 0x1.ffffffd0c621cp-2 : -0x1.ffffffd0c621cp-2;
  const double c2 = flip ? -0x1.55553e1068f19p-5 : 0x1.55553e1068f19p-5;
  const double c3 = flip ? 0x1.6c087e89a359dp-10 : -0x1.6c087e89a359dp-10;
  const double c4 = flip ? -0x1.99343027bf8c3p-16 : 0x1.99343027bf8c3p-16;
  const double s1 = -0x1.555545995a603p-3;
  const double s2 = 0x1.1107605230bc4p-7;
  const double s3 = -0x1.994eb3774cf24p-13;
  if ((n & 1) == 0) {
    double x3 = x * x2;
    double t1 = s2 + x2 * s3;
    double x7 = x3 * x2;
    double s = x + x3 * s1;
    return (float)(s + x7 * t1);
  } else {
    double x4 = x2 * x2;
    double t2 = c3 + x2 * c4;
    double t1 = c0 + x2 * c1;
    double x6 = x4 * x2;
    double c = t1 + x4 * c2;
    return (float)(c + x6 * t2);
  }
}

RGBL_HD double sc_reduce_fast(double x, int* np) {
  const double hpi_inv = 0x1.45F306DC9C883p+23;  // 2/pi * 2^24
  const double hpi = 0x1.921FB54442D18p0;
  double r = x * hpi_inv;
  int n = ((int32_t)r + 0x800000) >> 24;
  *np = n;
  return x - n * hpi;
}

// Valid for |y| < 120 (the reference only passes angle*pi/180 with angle in [0,360)).
RGBL_HD float glibc_sinf(float y) {
  double x = y;
  const uint32_t top = sc_abstop12(y);
  if (top < sc_abstop12(0x1.921FB6p-1f)) {
    double s = x * x;
    if (top < sc_abstop12(0x1p-12f)) return y;
    return sc_poly(x, s, 0, false);
  }
  int n;
  x = sc_reduce_fast(x, &n);
  const double sg = ((n & 3) == 1 || (n & 3) == 2) ? -1.0 : 1.0;
  return sc_poly(x * sg, x * x, n, (n & 2) != 0);
}

RGBL_HD float glibc_cosf(float y) {
  double x = y;
  const uint32_t top = sc_abstop12(y);
  if (top < sc_abstop12(0x1.921FB6p-1f)) {
    double x2 = x * x;
    if (top < sc_abstop12(0x1p-12f)) return 1.0f;
    return sc_poly(x, x2, 1, false);
  }
  int n;
  x = sc_reduce_fast(x, &n);
  const double sg = ((n & 3) == 1 || (n & 3) == 2) ? -1.0 : 1.0;
  return sc_poly(x * sg, x * x, n ^ 1, (n & 2) != 0);
}

}  // namespace rgbl
