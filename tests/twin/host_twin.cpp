// CPU twin of the HIP arithmetic library: the SAME headers compiled for the
// host with clang++ (no HIP), exposed through a tiny C ABI so that pytest can
// check the device algorithms against the golden fixtures without a GPU.
// Built with -DGS_FQ28_CHECK: every radix-2^28 multiplication asserts its limb
// contract and every lazy add/sub asserts that int32 limbs would not overflow.
// Test infrastructure only -- never loaded by the product.
#include <string.h>
#include <atomic>
#include <condition_variable>
#include <mutex>
#include <thread>
#include <vector>
#include "../../groth_sahai_rs_amd/csrc/gs_params_bls12_381.h"
#include "../../groth_sahai_rs_amd/csrc/gs_params_bn254.h"
#include "../../groth_sahai_rs_amd/csrc/gs_pairing.cuh"
#include "../../groth_sahai_rs_amd/csrc/gs_coop.cuh"
#include "../../groth_sahai_rs_amd/csrc/gs_wire.cuh"

namespace gs {
GS_ZERO_ONE(Bls12_381)
GS_ZERO_ONE(Bn254)
}
using namespace gs;

// The 3-lane cooperative exponentiation (gs_coop.cuh) on three host threads: the exchange policy is a
// mailbox with a barrier where the device uses wave shuffles; the lane code is the device's.
struct Barrier3 {
  std::mutex m;
  std::condition_variable cv;
  int waiting = 0, phase = 0;
  void wait() {
    std::unique_lock<std::mutex> l(m);
    int ph = phase;
    if (++waiting == 3) {
      waiting = 0;
      phase++;
      cv.notify_all();
    } else {
      cv.wait(l, [&] { return phase != ph; });
    }
  }
};
template <class C> struct CoopHost {
  Fp4<C>* slots;
  Barrier3* bar;
  Fp4<C> swap12(const Fp4<C>& mine, int j) {
    slots[j] = mine;
    bar->wait();
    Fp4<C> r = slots[j == 0 ? 0 : 3 - j];
    bar->wait();
    return r;
  }
  void gather(Fp4<C>* A, const Fp4<C>& mine, int j) {
    slots[j] = mine;
    bar->wait();
    for (int i = 0; i < 3; i++) A[i] = slots[i];
    bar->wait();
  }
};

template <class C> struct Twin {
  typedef Fq<C> F1;
  typedef Fp2<C> F2;
  static constexpr int NB = C::N * 4;  // boundary bytes of one Fq
  static F1 ld1(const uint8_t* p) { return fq_from_boundary<C>((const uint32_t*)p); }
  static void st1(uint8_t* p, const F1& a) { fq_to_boundary<C>((uint32_t*)p, a); }
  static Aff<F1> ldg1(const uint8_t* p) { return {ld1(p), ld1(p + NB)}; }
  static void stg1(uint8_t* p, const Aff<F1>& a) { st1(p, a.x); st1(p + NB, a.y); }
  static Aff<F2> ldg2(const uint8_t* p) { return {{ld1(p), ld1(p + NB)}, {ld1(p + 2 * NB), ld1(p + 3 * NB)}}; }
  static void stg2(uint8_t* p, const Aff<F2>& a) { st1(p, a.x.c0); st1(p + NB, a.x.c1); st1(p + 2 * NB, a.y.c0); st1(p + 3 * NB, a.y.c1); }

  static void fp_mul(const uint8_t* a, const uint8_t* b, uint8_t* o) { st1(o, mul(ld1(a), ld1(b))); }
  static void fp_inv(const uint8_t* a, uint8_t* o) { st1(o, inv(ld1(a))); }
  static void fp_addsub(const uint8_t* a, const uint8_t* b, uint8_t* o) {  // o = [a+b, a-b, -a, 8a - 5b (lazy chain)]
    F1 x = ld1(a), y = ld1(b);
    st1(o, add(x, y));
    st1(o + NB, sub(x, y));
    st1(o + 2 * NB, neg(x));
    F1 t = sub(norm(mul_small(x, 4)), y);   // 4x - y
    t = sub(dbl(t), mul_small(y, 3));       // 8x - 5y
    st1(o + 3 * NB, t);
  }
  static int fp_is_zero(const uint8_t* a, const uint8_t* b) {  // is_zero(a - b) through the lazy path
    F1 x = ld1(a), y = ld1(b);
    F1 d = sub(add(x, x), add(y, y));
    return is_zero(d) ? 1 : 0;
  }
  static void fr_mul(const uint32_t* a, const uint32_t* b, uint32_t* o) {
    Fr<C> x, y;
    memcpy(&x, a, sizeof x);
    memcpy(&y, b, sizeof y);
    Fr<C> r = mul(x, y);
    memcpy(o, &r, sizeof r);
  }
  static void g1_smul(const uint8_t* p, const uint32_t* k_mont, uint8_t* o) {
    Aff<F1> P = ldg1(p), R;
    Fr<C> k;
    memcpy(&k, k_mont, sizeof k);
    Jac<F1> J;
    jac_smul_any<C>(J, P, from_mont(k));
    jac_to_aff(R, J);
    stg1(o, R);
  }
  static void g2_smul(const uint8_t* p, const uint32_t* k_mont, uint8_t* o) {
    Aff<F2> P = ldg2(p), R;
    Fr<C> k;
    memcpy(&k, k_mont, sizeof k);
    Jac<F2> J;
    jac_smul_any<C>(J, P, from_mont(k));
    jac_to_aff(R, J);
    stg2(o, R);
  }
  // joint MSM over nt <= 8 terms (Straus), both groups
  static void g1_msm(int nt, const uint8_t* p, const uint32_t* k_mont, uint8_t* o) {
    Aff<F1> P[8], R;
    Fr<C> k[8];
    for (int i = 0; i < nt; i++) {
      P[i] = ldg1(p + i * 2 * NB);
      Fr<C> t;
      memcpy(&t, k_mont + i * 8, sizeof t);
      k[i] = from_mont(t);
    }
    Jac<F1> J;
    jac_msm_straus<C, F1, 8>(J, P, k, nt);
    jac_to_aff(R, J);
    stg1(o, R);
  }
  static void g2_msm(int nt, const uint8_t* p, const uint32_t* k_mont, uint8_t* o) {
    Aff<F2> P[8], R;
    Fr<C> k[8];
    for (int i = 0; i < nt; i++) {
      P[i] = ldg2(p + i * 4 * NB);
      Fr<C> t;
      memcpy(&t, k_mont + i * 8, sizeof t);
      k[i] = from_mont(t);
    }
    Jac<F2> J;
    jac_msm_straus<C, F2, 8>(J, P, k, nt);
    jac_to_aff(R, J);
    stg2(o, R);
  }
  static void g1_add(const uint8_t* p, const uint8_t* q, uint8_t* o) {
    Aff<F1> P = ldg1(p), Q = ldg1(q), R;
    Jac<F1> J, K;
    jac_from_aff(J, P);
    jac_from_aff(K, Q);
    jac_add(J, J, K);
    jac_to_aff(R, J);
    stg1(o, R);
  }
  static void g1_to_aff(const uint8_t* p, uint8_t* o) {  // Jacobian -> affine normalisation (one inversion)
    Aff<F1> P = ldg1(p), R;
    Jac<F1> J;
    jac_from_aff(J, P);
    jac_dbl(J, J);
    jac_to_aff(R, J);
    stg1(o, R);
  }
  static void g2_madd(const uint8_t* p, const uint8_t* q, uint8_t* o) {
    Aff<F2> P = ldg2(p), Q = ldg2(q), R;
    Jac<F2> J;
    jac_from_aff(J, P);
    jac_madd(J, J, Q);
    jac_to_aff(R, J);
    stg2(o, R);
  }
  // op: 0 mul, 1 sqr, 2 inv, 3 conj, 4 frob1, 5 frob2, 6 frob3, 7 cyclo_sqr
  static void fp12_op(int op, const uint8_t* a, const uint8_t* b, uint8_t* o) {
    Fp12<C> x, y, r;
    f12_from_boundary<C>(x, (const BFq<C>*)a);
    if (b) f12_from_boundary<C>(y, (const BFq<C>*)b);
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
    f12_to_boundary<C>((BFq<C>*)o, r);
  }
  // wire format of points (gs_wire.cuh): boundary affine <-> bytes
  static void wire_enc(int group, int compressed, const uint8_t* pt, uint8_t* out) {
    if (group == 1) wire_encode_point<C, F1>(out, ldg1(pt), compressed != 0);
    else wire_encode_point<C, F2>(out, ldg2(pt), compressed != 0);
  }
  static int wire_dec(int group, int compressed, int validate, const uint8_t* in, uint8_t* pt) {
    bool ok;
    if (group == 1) {
      Aff<F1> p;
      ok = wire_decode_point<C, F1>(p, in, compressed != 0, validate != 0);
      stg1(pt, p);
    } else {
      Aff<F2> p;
      ok = wire_decode_point<C, F2>(p, in, compressed != 0, validate != 0);
      stg2(pt, p);
    }
    return ok ? 1 : 0;
  }
  // what: 0 = f^x only, 1 = whole final exponentiation.  out = 3 GT values (one per lane; they must agree)
  static void coop(int what, const uint8_t* in, uint8_t* out) {
    Fp12<C> f;
    f12_from_boundary<C>(f, (const BFq<C>*)in);
    Fp4<C> slots[3];
    Barrier3 bar;
    Fp12<C> res[3];
    std::thread th[3];
    for (int j = 0; j < 3; j++)
      th[j] = std::thread([&, j] {
        CoopHost<C> xh{slots, &bar};
        ExpXCoop<C, CoopHost<C>> ex{j, &xh};
        if (what == 0) ex(res[j], f);
        else final_exp_with(res[j], f, ex);
      });
    for (int j = 0; j < 3; j++) th[j].join();
    for (int j = 0; j < 3; j++) f12_to_boundary<C>((BFq<C>*)out + 12 * j, res[j]);
  }
  static void exp_by_x(const uint8_t* in, uint8_t* out) {
    Fp12<C> f, r;
    f12_from_boundary<C>(f, (const BFq<C>*)in);
    f12_exp_by_x(r, f);
    f12_to_boundary<C>((BFq<C>*)out, r);
  }
  // Fq multiplications one device primitive executes (tools/count_fq_muls.py -> bench.py's ALU roofline).
  // g1/g2: nt affine points, k: nt Montgomery scalars.
  static long opcount(int op, int nt, const uint8_t* g1, const uint8_t* g2, const uint32_t* k_mont) {
    Aff<F1> P[8];
    Aff<F2> Q[8];
    Fr<C> k[8];
    for (int i = 0; i < nt && i < 8; i++) {
      P[i] = ldg1(g1 + i * 2 * NB);
      Q[i] = ldg2(g2 + i * 4 * NB);
      Fr<C> t;
      memcpy(&t, k_mont + i * 8, sizeof t);
      k[i] = from_mont(t);
    }
    Jac<F1> J1, K1;
    Jac<F2> J2, K2;
    jac_from_aff(J1, P[0]);
    jac_dbl(J1, J1);
    jac_from_aff(K1, P[1 % (nt ? nt : 1)]);
    jac_dbl(K1, K1);
    jac_madd(K1, K1, P[0]);
    jac_from_aff(J2, Q[0]);
    jac_dbl(J2, J2);
    jac_from_aff(K2, Q[1 % (nt ? nt : 1)]);
    jac_dbl(K2, K2);
    jac_madd(K2, K2, Q[0]);
    Fp12<C> f, g;
    {
      Proj2<C> T[8];
      bool live[8];
      multi_miller(f, P, Q, 1, T, live);
    }
    long c0 = fq28_mul_counter().load();
    long m0 = fq28_mad_counter().load();
    switch (op) {
      case 0: jac_smul_any<C>(J1, P[0], k[0]); break;
      case 1: jac_smul_any<C>(J2, Q[0], k[0]); break;
      case 2: if (nt <= 4) jac_msm_straus<C, F1, 4>(J1, P, k, nt); else jac_msm_straus<C, F1, 8>(J1, P, k, nt); break;
      case 3: if (nt <= 4) jac_msm_straus<C, F2, 4>(J2, Q, k, nt); else jac_msm_straus<C, F2, 8>(J2, Q, k, nt); break;
      case 17:    // G1 / G2 Straus lane: (nt & 15) bases, window width (nt >> 8) & 15, (nt >> 12) outputs per build
      case 18: {
        int n = nt & 15, w = (nt >> 8) & 15, mo = nt >> 12;
        if (op == 17) {
          static Aff<F1> at[8 * 16];
          static Jac<F1> jt1[8 * 16];
          F1 zb;
          if (w == 5) jac_straus_build<C, F1, 8, 5>(at, zb, P, n, jt1); else jac_straus_build<C, F1, 8, 4>(at, zb, P, n, jt1);
          for (int o = 0; o < mo; o++)
            if (w == 5) jac_straus_run<C, F1, 8, 5>(J1, k, n, at, zb); else jac_straus_run<C, F1, 8, 4>(J1, k, n, at, zb);
        } else {
          static Aff<F2> at[8 * 16];
          static Jac<F2> jt2[8 * 16];
          F2 zb;
          if (w == 5) jac_straus_build<C, F2, 8, 5>(at, zb, Q, n, jt2); else jac_straus_build<C, F2, 8, 4>(at, zb, Q, n, jt2);
          for (int o = 0; o < mo; o++)
            if (w == 5) jac_straus_run<C, F2, 8, 5>(J2, k, n, at, zb); else jac_straus_run<C, F2, 8, 4>(J2, k, n, at, zb);
        }
        break;
      }
      case 4: jac_madd(J1, J1, P[0]); break;
      case 5: jac_madd(J2, J2, Q[0]); break;
      case 6: jac_add(J1, J1, K1); break;
      case 7: jac_add(J2, J2, K2); break;
      case 8: {  // tail of k_red: one inversion for two points
        F1 zi = inv(mul(J1.z, K1.z));
        F1 a = mul(zi, K1.z), b = mul(zi, J1.z);
        Aff<F1> r0, r1;
        jac_to_aff_zinv(r0, J1, a);
        jac_to_aff_zinv(r1, K1, b);
        break;
      }
      case 9: {
        F2 zi = inv(mul(J2.z, K2.z));
        F2 a = mul(zi, K2.z), b = mul(zi, J2.z);
        Aff<F2> r0, r1;
        jac_to_aff_zinv(r0, J2, a);
        jac_to_aff_zinv(r1, K2, b);
        break;
      }
      case 10: {
        Proj2<C> T[8];
        bool live[8];
        multi_miller(g, P, Q, nt, T, live);
        break;
      }
      case 14: {  // twin Miller: nt (Q, P0, P1) triples, two accumulators
        Proj2<C> T[8];
        uint8_t live[8];
        Fp12<C> g1v;
        multi_miller2(g, g1v, P, P, Q, nt, T, live);
        break;
      }
      case 15:    // single Miller, nt pairs all reading line tables
      case 16: {  // twin Miller, nt triples all reading line tables
        constexpr int NLN = miller_line_count<C>();
        Line<C>* tabs = new Line<C>[8 * NLN];
        const Line<C>* fx[8];
        for (int i = 0; i < nt; i++) {
          miller_line_table<C>(tabs + i * NLN, Q[i]);
          fx[i] = tabs + i * NLN;
        }
        Proj2<C> T[8];
        c0 = fq28_mul_counter().load();
        m0 = fq28_mad_counter().load();
        if (op == 15) {
          bool live[8];
          multi_miller(g, P, Q, nt, T, live, fx);
        } else {
          uint8_t live[8];
          Fp12<C> g1v;
          multi_miller2(g, g1v, P, P, Q, nt, T, live, fx);
        }
        long v = fq28_mul_counter().load() - c0;
        last_mads() = fq28_mad_counter().load() - m0;
        delete[] tabs;
        return v;
      }
      case 11: f12_mul(g, f, f); break;
      case 12: final_exp(g, f); break;
      case 13: {  // whole 3-lane group (divide by 3 for one lane)
        uint8_t in[12 * NB], out[3 * 12 * NB];
        f12_to_boundary<C>((BFq<C>*)in, f);
        c0 = fq28_mul_counter().load();
        m0 = fq28_mad_counter().load();
        coop(1, in, out);
        last_mads() = fq28_mad_counter().load() - m0 - (3 * 12 + 12) * 2L * C::L * C::L;
        return fq28_mul_counter().load() - c0 - 3 * 12 - 12;  // minus the boundary conversions of this harness
      }
      default: break;
    }
    last_mads() = fq28_mad_counter().load() - m0;
    return fq28_mul_counter().load() - c0;
  }
  // multiply-add instructions the device executes for the primitive last counted by opcount (fq28_mad_counter)
  static long& last_mads() {
    static long v = 0;
    return v;
  }
  // the same with the pairs selected by `mask` reading a precomputed line table of their G2 argument
  static void multi_pairing_fixed(int np, const uint8_t* ps, const uint8_t* qs, unsigned mask, uint8_t* o, int twin_mode) {
    constexpr int NLN = miller_line_count<C>();
    Aff<F1>* P = new Aff<F1>[np];
    Aff<F2>* Q = new Aff<F2>[np];
    Proj2<C>* T = new Proj2<C>[np];
    Line<C>* tabs = new Line<C>[np * NLN];
    const Line<C>** fx = new const Line<C>*[np];
    for (int i = 0; i < np; i++) {
      P[i] = ldg1(ps + i * 2 * NB);
      Q[i] = ldg2(qs + i * 4 * NB);
      fx[i] = nullptr;
      if ((mask >> i) & 1) {
        miller_line_table<C>(tabs + i * NLN, Q[i]);
        fx[i] = tabs + i * NLN;
      }
    }
    Fp12<C> f, f1, e;
    if (twin_mode == 2) {
      // the lane-PAIR form of the twin loop (multi_miller_pair): two host threads stand for lanes 2i and 2i + 1, each
      // with one accumulator, the stepping triples dealt out alternately; lines cross through a two-slot exchange
      // with the same put / get discipline as the device's LDS slots.  Stepping triples come first, as the planner
      // orders a task's list.
      std::vector<int> ord;
      for (int i = 0; i < np; i++)
        if (!((mask >> i) & 1)) ord.push_back(i);
      const int nstep = (int)ord.size();
      for (int i = 0; i < np; i++)
        if ((mask >> i) & 1) ord.push_back(i);
      std::vector<Aff<F1>> Pp(np);
      std::vector<const Line<C>*> fxp(np);
      std::vector<Aff<F2>> qo[2];
      uint32_t qok = 0;
      for (int k = 0; k < np; k++) {
        Pp[k] = P[ord[k]];
        fxp[k] = fx[ord[k]];
        if (!aff_is_inf(Q[ord[k]])) qok |= 1u << k;
        if (k < nstep) qo[k & 1].push_back(Q[ord[k]]);
      }
      for (int a = 0; a < 2; a++) qo[a].resize((nstep + 1) / 2 + 1);
      struct Exchange {
        Line<C> slot[2];
        std::atomic<int> arrived{0}, phase{0};
        void barrier() {
          int ph = phase.load();
          if (arrived.fetch_add(1) == 1) {
            arrived.store(0);
            phase.store(ph + 1);
          } else {
            while (phase.load() == ph) std::this_thread::yield();
          }
        }
      } ex;
      struct Lane {
        Exchange* ex;
        int lane;
        void put(const Line<C>& m) {
          ex->barrier();  // the partner has taken the previous line
          ex->slot[lane] = m;
          ex->barrier();
        }
        Line<C> get() const { return ex->slot[lane ^ 1]; }
      };
      Fp12<C> acc[2];
      std::thread th[2];
      for (int a = 0; a < 2; a++)
        th[a] = std::thread([&, a] {
          Lane x{&ex, a};
          std::vector<Proj2<C>> ts((nstep + 1) / 2 + 1);
          multi_miller_pair(acc[a], a, Pp.data(), qo[a].data(), qok, nstep, np, ts.data(), fxp.data(), x);
        });
      for (int a = 0; a < 2; a++) th[a].join();
      f = acc[0];
      f1 = acc[1];
    } else if (twin_mode) {
      uint8_t* live = new uint8_t[np];
      multi_miller2(f, f1, P, P, Q, np, T, live, fx);
      delete[] live;
    } else {
      bool* live = new bool[np];
      multi_miller(f, P, Q, np, T, live, fx);
      delete[] live;
    }
    final_exp(e, f);
    f12_to_boundary<C>((BFq<C>*)o, e);
    if (twin_mode) {
      final_exp(e, f1);
      f12_to_boundary<C>((BFq<C>*)o + 12, e);
    }
    delete[] P; delete[] Q; delete[] T; delete[] tabs; delete[] fx;
  }
  static void multi_pairing(int np, const uint8_t* ps, const uint8_t* qs, uint8_t* o, int do_fe) {
    Aff<F1>* P = new Aff<F1>[np];
    Aff<F2>* Q = new Aff<F2>[np];
    Proj2<C>* T = new Proj2<C>[np];
    bool* live = new bool[np];
    for (int i = 0; i < np; i++) {
      P[i] = ldg1(ps + i * 2 * NB);
      Q[i] = ldg2(qs + i * 4 * NB);
    }
    Fp12<C> f, e;
    multi_miller(f, P, Q, np, T, live);
    if (do_fe) final_exp(e, f); else e = f;
    f12_to_boundary<C>((BFq<C>*)o, e);
    delete[] P; delete[] Q; delete[] T; delete[] live;
  }
};

extern "C" long twin_fq_mul_count(int reset) {
  long v = fq28_mul_counter().load();
  if (reset) fq28_mul_counter().store(0);
  return v;
}

#define EXPORT(SUF, CURVE)                                                                                        \
  extern "C" {                                                                                                    \
  void twin_fp_mul_##SUF(const uint8_t* a, const uint8_t* b, uint8_t* o) { Twin<CURVE>::fp_mul(a, b, o); }       \
  void twin_fp_inv_##SUF(const uint8_t* a, uint8_t* o) { Twin<CURVE>::fp_inv(a, o); }                            \
  void twin_fp_addsub_##SUF(const uint8_t* a, const uint8_t* b, uint8_t* o) { Twin<CURVE>::fp_addsub(a, b, o); } \
  int twin_fp_is_zero_##SUF(const uint8_t* a, const uint8_t* b) { return Twin<CURVE>::fp_is_zero(a, b); }        \
  void twin_fr_mul_##SUF(const uint32_t* a, const uint32_t* b, uint32_t* o) { Twin<CURVE>::fr_mul(a, b, o); }    \
  void twin_g1_smul_##SUF(const uint8_t* p, const uint32_t* k, uint8_t* o) { Twin<CURVE>::g1_smul(p, k, o); }    \
  void twin_g2_smul_##SUF(const uint8_t* p, const uint32_t* k, uint8_t* o) { Twin<CURVE>::g2_smul(p, k, o); }    \
  void twin_g1_msm_##SUF(int nt, const uint8_t* p, const uint32_t* k, uint8_t* o) { Twin<CURVE>::g1_msm(nt, p, k, o); } \
  void twin_g2_msm_##SUF(int nt, const uint8_t* p, const uint32_t* k, uint8_t* o) { Twin<CURVE>::g2_msm(nt, p, k, o); } \
  void twin_g1_add_##SUF(const uint8_t* p, const uint8_t* q, uint8_t* o) { Twin<CURVE>::g1_add(p, q, o); }       \
  void twin_g1_to_aff_##SUF(const uint8_t* p, uint8_t* o) { Twin<CURVE>::g1_to_aff(p, o); }                       \
  void twin_g2_madd_##SUF(const uint8_t* p, const uint8_t* q, uint8_t* o) { Twin<CURVE>::g2_madd(p, q, o); }     \
  void twin_fp12_op_##SUF(int op, const uint8_t* a, const uint8_t* b, uint8_t* o) {                              \
    Twin<CURVE>::fp12_op(op, a, b, o);                                                                            \
  }                                                                                                               \
  long twin_opcount_##SUF(int op, int nt, const uint8_t* g1, const uint8_t* g2, const uint32_t* k) {              \
    return Twin<CURVE>::opcount(op, nt, g1, g2, k);                                                                \
  }                                                                                                               \
  long twin_last_mads_##SUF() { return Twin<CURVE>::last_mads(); }                                                 \
  void twin_wire_enc_##SUF(int g, int c, const uint8_t* pt, uint8_t* out) { Twin<CURVE>::wire_enc(g, c, pt, out); }  \
  int twin_wire_dec_##SUF(int g, int c, int v, const uint8_t* in, uint8_t* pt) {                                  \
    return Twin<CURVE>::wire_dec(g, c, v, in, pt);                                                                 \
  }                                                                                                               \
  void twin_coop_##SUF(int what, const uint8_t* in, uint8_t* out) { Twin<CURVE>::coop(what, in, out); }           \
  void twin_exp_by_x_##SUF(const uint8_t* in, uint8_t* out) { Twin<CURVE>::exp_by_x(in, out); }                   \
  void twin_multi_pairing_fixed_##SUF(int np, const uint8_t* ps, const uint8_t* qs, unsigned mask, uint8_t* o,     \
                                      int twin_mode) {                                                           \
    Twin<CURVE>::multi_pairing_fixed(np, ps, qs, mask, o, twin_mode);                                              \
  }                                                                                                               \
  void twin_multi_pairing_##SUF(int np, const uint8_t* ps, const uint8_t* qs, uint8_t* o, int fe) {              \
    Twin<CURVE>::multi_pairing(np, ps, qs, o, fe);                                                                \
  }                                                                                                               \
  }
EXPORT(bls12_381, Bls12_381)
EXPORT(bn254, Bn254)
