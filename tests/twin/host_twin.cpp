// CPU twin of the HIP arithmetic library: the SAME headers compiled for the
// host with clang++ (no HIP), exposed through a tiny C ABI so that pytest can
// check the device algorithms against the golden fixtures without a GPU.
// Test infrastructure only -- never loaded by the product.
#include <string.h>
#include "../../groth_sahai_rs_amd/csrc/gs_params_bls12_381.h"
#include "../../groth_sahai_rs_amd/csrc/gs_params_bn254.h"
#include "../../groth_sahai_rs_amd/csrc/gs_pairing.cuh"

namespace gs {
GS_ZERO_ONE(Bls12_381)
GS_ZERO_ONE(Bn254)
}
using namespace gs;

template <class C> struct Twin {
  typedef Fq<C> F1;
  typedef Fp2<C> F2;
  static void fp_mul(const uint32_t* a, const uint32_t* b, uint32_t* o) {
    F1 x, y;
    memcpy(&x, a, sizeof x);
    memcpy(&y, b, sizeof y);
    F1 r = mul(x, y);
    memcpy(o, &r, sizeof r);
  }
  static void fp_inv(const uint32_t* a, uint32_t* o) {
    F1 x;
    memcpy(&x, a, sizeof x);
    F1 r = inv(x);
    memcpy(o, &r, sizeof r);
  }
  static void fp_addsub(const uint32_t* a, const uint32_t* b, uint32_t* o) {  // o = [a+b, a-b, -a, a/2]
    F1 x, y;
    memcpy(&x, a, sizeof x);
    memcpy(&y, b, sizeof y);
    F1 r[4] = {add(x, y), sub(x, y), neg(x), half(x)};
    memcpy(o, r, sizeof r);
  }
  static void fr_mul(const uint32_t* a, const uint32_t* b, uint32_t* o) {
    Fr<C> x, y;
    memcpy(&x, a, sizeof x);
    memcpy(&y, b, sizeof y);
    Fr<C> r = mul(x, y);
    memcpy(o, &r, sizeof r);
  }
  static void g1_smul(const uint32_t* p, const uint32_t* k_mont, uint32_t* o) {
    Aff<F1> P, R;
    Fr<C> k;
    memcpy(&P, p, sizeof P);
    memcpy(&k, k_mont, sizeof k);
    Jac<F1> J;
    jac_smul(J, P, from_mont(k));
    jac_to_aff(R, J);
    memcpy(o, &R, sizeof R);
  }
  static void g2_smul(const uint32_t* p, const uint32_t* k_mont, uint32_t* o) {
    Aff<F2> P, R;
    Fr<C> k;
    memcpy(&P, p, sizeof P);
    memcpy(&k, k_mont, sizeof k);
    Jac<F2> J;
    jac_smul(J, P, from_mont(k));
    jac_to_aff(R, J);
    memcpy(o, &R, sizeof R);
  }
  static void g1_add(const uint32_t* p, const uint32_t* q, uint32_t* o) {
    Aff<F1> P, Q, R;
    memcpy(&P, p, sizeof P);
    memcpy(&Q, q, sizeof Q);
    Jac<F1> J, K;
    jac_from_aff(J, P);
    jac_from_aff(K, Q);
    jac_add(J, J, K);
    jac_to_aff(R, J);
    memcpy(o, &R, sizeof R);
  }
  // op: 0 mul, 1 sqr, 2 inv, 3 conj, 4 frob1, 5 frob2, 6 frob3, 7 cyclo_sqr
  static void fp12_op(int op, const uint32_t* a, const uint32_t* b, uint32_t* o) {
    Fp12<C> x, y, r;
    memcpy(&x, a, sizeof x);
    if (b) memcpy(&y, b, sizeof y);
    switch (op) {
      case 0: f12_mul(r, x, y); break;
      case 1: f12_sqr(r, x); break;
      case 2: f12_inv(r, x); break;
      case 3: f12_conj(r, x); break;
      case 4: f12_frob(r, x, 1); break;
      case 5: f12_frob(r, x, 2); break;
      case 6: f12_frob(r, x, 3); break;
      case 7: f12_cyclo_sqr(r, x); break;
      default: r = x;
    }
    memcpy(o, &r, sizeof r);
  }
  static void multi_pairing(int np, const uint32_t* ps, const uint32_t* qs, uint32_t* o, int do_fe) {
    Aff<F1>* P = new Aff<F1>[np];
    Aff<F2>* Q = new Aff<F2>[np];
    Proj2<C>* T = new Proj2<C>[np];
    bool* live = new bool[np];
    memcpy(P, ps, sizeof(Aff<F1>) * np);
    memcpy(Q, qs, sizeof(Aff<F2>) * np);
    Fp12<C> f, e;
    multi_miller(f, P, Q, np, T, live);
    if (do_fe) final_exp(e, f); else e = f;
    memcpy(o, &e, sizeof e);
    delete[] P; delete[] Q; delete[] T; delete[] live;
  }
  static void final_exp_only(const uint32_t* a, uint32_t* o) {
    Fp12<C> f, e;
    memcpy(&f, a, sizeof f);
    final_exp(e, f);
    memcpy(o, &e, sizeof e);
  }
};

#define EXPORT(SUF, CURVE)                                                                                        \
  extern "C" {                                                                                                    \
  void twin_fp_mul_##SUF(const uint32_t* a, const uint32_t* b, uint32_t* o) { Twin<CURVE>::fp_mul(a, b, o); }    \
  void twin_fp_inv_##SUF(const uint32_t* a, uint32_t* o) { Twin<CURVE>::fp_inv(a, o); }                          \
  void twin_fp_addsub_##SUF(const uint32_t* a, const uint32_t* b, uint32_t* o) { Twin<CURVE>::fp_addsub(a, b, o); } \
  void twin_fr_mul_##SUF(const uint32_t* a, const uint32_t* b, uint32_t* o) { Twin<CURVE>::fr_mul(a, b, o); }    \
  void twin_g1_smul_##SUF(const uint32_t* p, const uint32_t* k, uint32_t* o) { Twin<CURVE>::g1_smul(p, k, o); }  \
  void twin_g2_smul_##SUF(const uint32_t* p, const uint32_t* k, uint32_t* o) { Twin<CURVE>::g2_smul(p, k, o); }  \
  void twin_g1_add_##SUF(const uint32_t* p, const uint32_t* q, uint32_t* o) { Twin<CURVE>::g1_add(p, q, o); }    \
  void twin_fp12_op_##SUF(int op, const uint32_t* a, const uint32_t* b, uint32_t* o) {                           \
    Twin<CURVE>::fp12_op(op, a, b, o);                                                                            \
  }                                                                                                               \
  void twin_multi_pairing_##SUF(int np, const uint32_t* ps, const uint32_t* qs, uint32_t* o, int fe) {           \
    Twin<CURVE>::multi_pairing(np, ps, qs, o, fe);                                                                \
  }                                                                                                               \
  void twin_final_exp_##SUF(const uint32_t* a, uint32_t* o) { Twin<CURVE>::final_exp_only(a, o); }               \
  }
EXPORT(bls12_381, Bls12_381)
EXPORT(bn254, Bn254)
