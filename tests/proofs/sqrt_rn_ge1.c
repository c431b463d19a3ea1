/* Exhaustive proof of the square-root correction in k_post (emap_kernels.hip: sqrt_rn_ge1): for EVERY float x in [1, +inf] and for
 * every start value within one ulp of the correctly rounded root (what v_sqrt_f32 guarantees), the two-residual correction returns
 * the correctly rounded root.  Prints "n=<cases> bad=<failures>".  Test infrastructure (tests/test_proofs.py). */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
static float asf(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
static uint32_t asu(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
int main(void) {
  unsigned long bad = 0, n = 0;
#pragma omp parallel for reduction(+ : bad, n) schedule(static)
  for (long u = 0x3f800000L; u <= 0x7f800000L; ++u) {
    const float x = asf((uint32_t)u), ref = sqrtf(x);
    for (int d = -1; d <= 1; ++d) {
      float s = isinf(x) ? ref : asf(asu(ref) + (uint32_t)d);
      const float sd = asf(asu(s) - 1u), su = asf(asu(s) + 1u);
      const float rd = fmaf(-sd, s, x), ru = fmaf(-su, s, x);
      s = rd <= 0.0f ? sd : s;
      s = ru > 0.0f ? su : s;
      ++n;
      if (asu(s) != asu(ref)) ++bad;
    }
  }
  printf("n=%lu bad=%lu\n", n, bad);
  return bad != 0;
}
