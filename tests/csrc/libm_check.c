/* Host-side check of ltransv.2b_amd/csrc/lt_libm.h: the same source the kernels compile, built for
 * the host with gcc, against the host libm (called through volatile pointers so that the compiler
 * neither folds the calls nor merges sin + cos into sincos).  Prints "<name> <mismatches> <samples>"
 * per function; tests/test_libm_host.py asserts the mismatch counts are 0. */
#define _GNU_SOURCE
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include "lt_libm.h"

static double (*volatile p_exp)(double) = exp;
static double (*volatile p_log)(double) = log;
static double (*volatile p_log10)(double) = log10;
static double (*volatile p_cos)(double) = cos;
static double (*volatile p_sin)(double) = sin;
static double (*volatile p_atan)(double) = atan;
static double (*volatile p_pow)(double, double) = pow;
static void (*volatile p_sincos)(double, double *, double *) = sincos;

static int same(double a, double b) { return lt_lm_asu(a) == lt_lm_asu(b); }
static double urand(double lo, double hi) { return lo + (hi - lo) * drand48(); }
static double lrand(double lo, double hi) { return exp(urand(log(lo), log(hi))); }

int main(int argc, char **argv) {
  const long n = (argc > 1) ? atol(argv[1]) : 1000000;
  long bad;
  srand48(12345);
  bad = 0;
  for (long i = 0; i < n; i++) {
    const double x = (i % 3 == 0) ? -urand(0.0, 170.0) : (i % 3 == 1) ? urand(-512.0, 512.0) : -lrand(1e-30, 1100.0);
    if (fabs(x) < 512.0 && !same(exp_core(x), p_exp(x))) bad++;
    if (x <= 0.0 && !same(lt_exp(x), p_exp(x))) bad++;
  }
  printf("exp %ld %ld\n", bad, n);
  bad = 0;
  for (long i = 0; i < n; i++) {
    const double x = (i % 3 == 0) ? urand(0.0, 1.0) : (i % 3 == 1) ? urand(0.9, 1.1) : lrand(1e-300, 1e300);
    if (x > 0.0 && !same(lt_log(x), p_log(x))) bad++;
  }
  printf("log %ld %ld\n", bad, n);
  bad = 0;
  for (long i = 0; i < n; i++) {
    const double x = (i % 2 == 0) ? lrand(1e-3, 1e5) : lrand(1e-300, 1e300);
    if (!same(lt_log10(x), p_log10(x))) bad++;
  }
  printf("log10 %ld %ld\n", bad, n);
  bad = 0;
  for (long i = 0; i < n; i++) {
    const double x = (i % 3 == 0) ? urand(4e-6, 1e-3) : (i % 3 == 1) ? urand(0.0, 500.0) : lrand(1e-300, 1e300);
    if (!same(lt_pow_half(x), p_pow(x, 0.5))) bad++;
  }
  if (!same(lt_pow_half(0.0), p_pow(0.0, 0.5)) || !same(lt_pow_half(1.0), p_pow(1.0, 0.5))) bad++;
  printf("pow_half %ld %ld\n", bad, n);
  {
    long badc = 0, bads = 0, badsc = 0;
    for (long i = 0; i < n; i++) {
      double x = (i % 4 == 0) ? urand(0.0, 6.283185307179586) : (i % 4 == 1) ? urand(-0.3, 0.3)
               : (i % 4 == 2) ? urand(-10.0, 10.0) : lrand(1e-9, 1e8);
      double s, c, s2, c2;
      if (i & 8) x = -x;
      if (!same(lt_cos(x), p_cos(x))) badc++;
      if (!same(lt_sin(x), p_sin(x))) bads++;
      lt_sincos(x, &s, &c);
      p_sincos(x, &s2, &c2);
      if (!same(s, s2) || !same(c, c2)) badsc++;
    }
    printf("cos %ld %ld\nsin %ld %ld\nsincos %ld %ld\n", badc, n, bads, n, badsc, n);
  }
  bad = 0;
  for (long i = 0; i < n; i++) {
    const double x = (i % 4 == 0) ? urand(-1.0, 1.0) : (i % 4 == 1) ? lrand(1e-13, 1e5) : (i % 4 == 2) ? urand(-20.0, 20.0)
                   : -lrand(1e-3, 1e17);
    if (!same(lt_atan(x), p_atan(x))) bad++;
  }
  printf("atan %ld %ld\n", bad, n);
  return 0;
}
