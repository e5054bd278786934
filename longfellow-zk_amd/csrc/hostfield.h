// hostfield.h -- host-side scalar field helpers shared by the C++ host loops (sumcheck layer, ZK prover).
#pragma once
#include "ctx.h"
#include "fs_crypto.h"

// host GF(2^128) product: PCLMULQDQ when the CPU has it (fs_crypto.cc), else the portable product of fields.h
static inline elt_t h_gf_mul(elt_t a, elt_t b) {
  uint64_t x[2] = {a.lo, a.hi}, y[2] = {b.lo, b.hi}, o[2];
  if (fs_gf128_mul(x, y, o)) return elt_t{o[0], o[1]};
  return gf_mul(a, b);
}

struct HostField {
  int field;
  elt_t one, pts[3], invden[3];
  elt_t add(elt_t a, elt_t b) const { return field == LFGPU_FIELD_GF2_128 ? gf_add(a, b) : fp_add(a, b); }
  elt_t sub(elt_t a, elt_t b) const { return field == LFGPU_FIELD_GF2_128 ? gf_add(a, b) : fp_sub(a, b); }
  elt_t mul(elt_t a, elt_t b) const { return field == LFGPU_FIELD_GF2_128 ? h_gf_mul(a, b) : fp_mul(a, b); }
  elt_t inv(elt_t a) const { return field == LFGPU_FIELD_GF2_128 ? h_gf_inv(a) : h_fp_inv(a); }
  explicit HostField(lfgpu_ctx* c, int f) : field(f) {
    if (f == LFGPU_FIELD_GF2_128) {  // poly_evaluation_points_ = 0, 1, g (lib/gf2k/gf2_128.h:121-127)
      one = elt_t{1, 0};
      pts[0] = elt_t{0, 0};
      pts[1] = one;
      pts[2] = lf_gf_ctx(c, 4)->g;
    } else {  // 0, 1, 2 (lib/algebra/fp_generic.h:114-121)
      one = h_fp_of_scalar(1);
      pts[0] = h_fp_of_scalar(0);
      pts[1] = one;
      pts[2] = h_fp_of_scalar(2);
    }
    for (int i = 0; i < 3; ++i) {
      elt_t d = one;
      for (int j = 0; j < 3; ++j)
        if (j != i) d = mul(d, sub(pts[i], pts[j]));
      invden[i] = inv(d);
    }
  }
  // Poly<3>::eval_monomial (lib/algebra/poly.h:100-108)
  elt_t eval_monomial(const elt_t coef[3], elt_t x) const { return add(mul(add(mul(coef[2], x), coef[1]), x), coef[0]); }
  // value at x of the quadratic through (pts[i], ev[i]) = Poly<3>::eval_lagrange (poly.h:72-98)
  elt_t eval_lagrange(const elt_t ev[3], elt_t x) const {
    elt_t acc{0, 0};
    if (field != LFGPU_FIELD_GF2_128) acc = pts[0];
    for (int i = 0; i < 3; ++i) {
      elt_t num = one;
      for (int j = 0; j < 3; ++j)
        if (j != i) num = mul(num, sub(x, pts[j]));
      acc = add(acc, mul(ev[i], mul(num, invden[i])));
    }
    return acc;
  }
};
