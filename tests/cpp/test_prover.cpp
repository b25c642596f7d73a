// The reference's integration tests (tests/prover.rs:25-172; prove.rs:510-589; commit.rs:440-548)
// restated in C++ against include/gs_amd.hpp, run on the GPU through the C ABI.
// Input: a case blob written by tests/test_gpu_cpp_host.py from tests/golden/*.json:
//   u32 curve, type, m, n; then length-prefixed (u64) sections
//   u0 u1 v0 v1 g1 g2 gt X Y A B Gamma target R S T xcoms ycoms pi theta
//   + wire-format expectations from oracle/gs_wire_oracle.py: Commit1 (compressed), EquProof (compressed),
//     the equation (uncompressed), the CRS (compressed)
// Exit code 0 and "OK <checks>" on success.
#include <sys/mman.h>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iostream>

#include "gs_amd.hpp"

using namespace gs_amd;

static int checks = 0;
#define CHECK(c)                                                   \
  do {                                                             \
    if (!(c)) {                                                    \
      std::fprintf(stderr, "FAIL %s:%d %s\n", __FILE__, __LINE__, #c); \
      std::exit(1);                                                \
    }                                                              \
    checks++;                                                      \
  } while (0)

struct ReplayRng {  // stands in for `&mut CR: Rng`: hands out the recorded draws in order
  std::vector<Fr> q;
  size_t i = 0;
  Fr fr() {
    if (i >= q.size()) {
      std::fprintf(stderr, "rng exhausted\n");
      std::exit(1);
    }
    return q[i++];
  }
};

static Bytes section(std::ifstream& f) {
  uint64_t n = 0;
  f.read((char*)&n, 8);
  Bytes b(n);
  f.read((char*)b.data(), n);
  if (!f) {
    std::fprintf(stderr, "short case file\n");
    std::exit(2);
  }
  return b;
}

// which variable group of a Statement a witness list belongs to, by its element type
static void put_x(StatementVars& v, const std::vector<G1Affine>& x) { v.xg = x; }
static void put_x(StatementVars& v, const std::vector<Fr>& x) { v.xs = x; }
static void put_y(StatementVars& v, const std::vector<G2Affine>& y) { v.yg = y; }
static void put_y(StatementVars& v, const std::vector<Fr>& y) { v.ys = y; }
static const Commit1& com_x(const StatementProof& p, const std::vector<G1Affine>&) { return p.com_xg; }
static const Commit1& com_x(const StatementProof& p, const std::vector<Fr>&) { return p.com_xs; }
static const Commit2& com_y(const StatementProof& p, const std::vector<G2Affine>&) { return p.com_yg; }
static const Commit2& com_y(const StatementProof& p, const std::vector<Fr>&) { return p.com_ys; }
// (only reached for PPE cases, where the X witnesses ARE G1 points; the other overloads keep the template compiling)
static G1Affine as_g1(const G1Affine& p) { return p; }
static G1Affine as_g1(const Fr&) { return G1Affine{}; }
static std::vector<G1Affine> as_g1v(const std::vector<G1Affine>& v) { return v; }
static std::vector<G1Affine> as_g1v(const std::vector<Fr>&) { return {}; }

template <class A1, class A2, class AT, EquType TY>
static void run(const CRS& crs, uint32_t m, uint32_t n, const Bytes& X, const Bytes& Y, const Bytes& A, const Bytes& B,
                const Bytes& G, const Bytes& tgt, const Bytes& R, const Bytes& S, const Bytes& T, const Bytes& xc,
                const Bytes& yc, const Bytes& pi, const Bytes& th, const Bytes& w_xcoms, const Bytes& w_proof,
                const Bytes& w_equ, const Bytes& w_crs, int curve) {
  using Equ = Equation<A1, A2, AT, TY>;
  size_t fr = crs.ctx->sz[1];
  Equ equ;
  equ.a_consts = split<A1>(A, n);
  equ.b_consts = split<A2>(B, m);
  auto gflat = split<Fr>(G, (size_t)m * n);
  equ.gamma.resize(m);
  for (uint32_t i = 0; i < m; i++) equ.gamma[i].assign(gflat.begin() + i * n, gflat.begin() + (i + 1) * n);
  equ.target.v = tgt;
  auto xvars = split<A1>(X, m);
  auto yvars = split<A2>(Y, n);
  auto rng_of = [&](std::initializer_list<const Bytes*> parts) {
    ReplayRng r;
    for (const Bytes* p : parts) {
      auto v = split<Fr>(*p, p->size() / fr);
      r.q.insert(r.q.end(), v.begin(), v.end());
    }
    return r;
  };

  // tests/prover.rs: verify(commit_and_prove(..)) and bit-exact outputs for the recorded draws
  ReplayRng rng = rng_of({&R, &S, &T});
  CProof proof = equ.commit_and_prove(xvars, yvars, crs, rng);
  CHECK(rng.i == rng.q.size());  // draw order R, S, T and nothing else
  CHECK(proof.equ_proofs.size() == 1);
  CHECK(proof.equ_proofs[0].equ_type == TY && equ.get_type() == TY);  // prove.rs:510-536
  CHECK(proof.equ_proofs[0].pi.size() == Equ::KX && proof.equ_proofs[0].theta.size() == Equ::KY);
  CHECK(cat(proof.xcoms.coms) == xc);
  CHECK(cat(proof.ycoms.coms) == yc);
  CHECK(cat(proof.equ_proofs[0].pi) == pi);
  CHECK(cat(proof.equ_proofs[0].theta) == th);
  CHECK(cat(proof.xcoms.rand) == R && cat(proof.ycoms.rand) == S && cat(proof.equ_proofs[0].rand) == T);
  CHECK(equ.verify(proof, crs));

  // prove.rs:538-589: commit_and_prove == batch commits + prove under re-synchronised RNGs
  ReplayRng rng2 = rng_of({&R, &S, &T});
  Commit1 xcoms = batch_commit_x(xvars, crs, rng2);
  Commit2 ycoms = batch_commit_y(yvars, crs, rng2);
  CHECK(xcoms == proof.xcoms && ycoms == proof.ycoms);
  EquProof pf = equ.prove(xvars, yvars, xcoms, ycoms, crs, rng2);
  CHECK(cat(pf.pi) == pi && cat(pf.theta) == th);

  // commit.rs:440-548: batch commit == sequence of single commits under a synchronised RNG
  {
    ReplayRng r1 = rng_of({&R});
    Commit1 acc;
    for (const auto& x : xvars) {
      Commit1 one = batch_commit_x(std::vector<A1>{x}, crs, r1);
      acc.append(one);
      CHECK(one.coms.empty() && one.rand.empty());
    }
    CHECK(acc == xcoms);
    ReplayRng r2 = rng_of({&S});
    Commit2 acc2;
    for (const auto& y : yvars) {
      Commit2 one = batch_commit_y(std::vector<A2>{y}, crs, r2);
      acc2.append(one);
    }
    CHECK(acc2 == ycoms);
  }

  // wire format (statement.rs:215-390, commit.rs:300-340, prove.rs:600-640: round trips; bytes = the oracle's)
  {
    CHECK(serialize_compressed(proof.xcoms, crs) == w_xcoms);
    CHECK(serialize_compressed(proof.equ_proofs[0], crs) == w_proof);
    CHECK(serialize_uncompressed(equ, crs) == w_equ);
    CHECK(serialize_compressed(crs, crs) == w_crs);
    Commit1 xc2 = deserialize_compressed<Commit1>(w_xcoms, crs);
    CHECK(xc2 == proof.xcoms);
    Commit2 yc2 = deserialize_uncompressed<Commit2>(serialize_uncompressed(proof.ycoms, crs), crs);
    CHECK(yc2 == proof.ycoms);
    EquProof pf2 = deserialize_compressed<EquProof>(w_proof, crs);
    CHECK(pf2.equ_type == TY && cat(pf2.pi) == pi && cat(pf2.theta) == th && cat(pf2.rand) == T);
    Equ equ2 = deserialize_uncompressed<Equ>(w_equ, crs);
    CHECK(serialize_uncompressed(equ2, crs) == w_equ);
    CRS crs2 = deserialize_crs(w_crs, true, curve);
    CHECK(equ2.verify(CProof{xc2, yc2, {pf2}}, crs2));
    bool threw = false;
    try {
      Bytes bad = w_proof;
      bad.pop_back();
      deserialize_compressed<EquProof>(bad, crs);
    } catch (const SerializationError&) {
      threw = true;
    }
    CHECK(threw);
    threw = false;
    try {
      Bytes bad = w_xcoms;
      // first point of the first commitment: BLS12-381 loses its compression flag, BN254 gets both flags at once
      if (curve == GS_CURVE_BLS12_381) bad[8] &= 0x7F; else bad[8 + crs.ctx->sz[0] - 1] = 0xFF;
      deserialize_compressed<Commit1>(bad, crs);
    } catch (const SerializationError&) {
      threw = true;
    }
    CHECK(threw);
  }

  // negatives: a tampered proof element, commitment and target must be rejected
  {
    CProof bad = proof;
    bad.equ_proofs[0].pi[0] = bad.equ_proofs[0].pi[Equ::KX - 1 ? 1 : 0];
    if (Equ::KX > 1) CHECK(!equ.verify(bad, crs));
    bad = proof;
    std::swap(bad.xcoms.coms[0], bad.xcoms.coms[m - 1]);
    if (m > 1 && !(proof.xcoms.coms[0] == proof.xcoms.coms[m - 1])) CHECK(!equ.verify(bad, crs));
    Equ other = equ;
    other.gamma[0][0] = gflat[0].v == Bytes(fr, 0) ? rng_of({&R}).fr() : Fr{Bytes(fr, 0)};
    CHECK(!other.verify(proof, crs));
  }

  // page-locked caller memory: the region registers and releases; registering it twice, or releasing it twice, is refused
  {
    // (a mapping of its own, not allocator heap: include/gs_amd.h, gs_host_register)
    const size_t len = 1 << 16;
    void* buf = mmap(nullptr, len, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
    CHECK(buf != MAP_FAILED);
    {
      PinnedRegion pr(*crs.ctx, buf, len);
      CHECK(gs_host_register(crs.ctx->c, buf, len) != GS_OK);
    }
    CHECK(gs_host_unregister(crs.ctx->c, buf) != GS_OK);
    munmap(buf, len);
  }

  // the several-GPU entry (gs_ctx_create_multi; one device on this box): a batch of three proofs of the same
  // statement, the middle one tampered -- verdicts per equation as Verifiable::verify gives them
  {
    MultiCtx mc(curve, {0});
    mc.set_crs(crs);
    CProof bad = proof;
    Equ other = equ;
    other.gamma[0][0] = gflat[0].v == Bytes(fr, 0) ? rng_of({&R}).fr() : Fr{Bytes(fr, 0)};
    std::vector<bool> ok = mc.verify_batch(std::vector<Equ>{equ, other, equ}, std::vector<CProof>{proof, proof, proof});
    CHECK(ok.size() == 3 && ok[0] && !ok[1] && ok[2]);
    bool threw = false;
    try {
      CProof shortp = proof;
      shortp.equ_proofs[0].pi.pop_back();
      mc.verify_batch(std::vector<Equ>{equ, equ}, std::vector<CProof>{proof, shortp});
    } catch (const Panic&) {
      threw = true;
    }
    CHECK(threw);
  }

  // a Statement (statement.rs:24-28,109): equations over ONE list of variables, committed once.  Three equations of
  // this case's type (the middle one with another Gamma) and -- for PPE cases -- an MSMEG1 over the same G1 variables
  // and two fresh scalar variables: the statement's proofs equal Provable::prove per equation against the shared
  // commitments under the same draws, its verdicts equal Verifiable::verify per equation, the first equation's proof
  // is the golden one.
  {
    Equ other = equ;
    other.gamma[0][0] = gflat[0].v == Bytes(fr, 0) ? rng_of({&R}).fr() : Fr{Bytes(fr, 0)};
    Statement st;
    st.push(equ);
    st.push(other);
    st.push(equ);
    StatementVars v;
    put_x(v, xvars);
    put_y(v, yvars);
    constexpr bool mix = TY == EquType::PairingProduct;
    MSMEG1 e1;
    std::vector<Fr> ys2;
    if (mix) {
      ReplayRng src = rng_of({&S, &R, &T});  // any scalars will do for the extra equation
      ys2 = {src.fr(), src.fr()};
      v.ys = ys2;
      e1.a_consts = {as_g1(xvars[0]), as_g1(xvars[m - 1])};
      for (uint32_t i = 0; i < m; i++) e1.b_consts.push_back(src.fr());
      e1.gamma.assign(m, std::vector<Fr>{src.fr(), src.fr()});
      e1.target = as_g1(xvars[0]);
      st.push(e1);
    }
    // draw order of the statement: commit randomness of xg, yg, xs, ys, then T per equation
    Bytes S2 = cat(std::vector<Fr>(ys2.size(), rng_of({&T}).fr()));  // randomness of the two extra scalar variables
    ReplayRng rs = TY == EquType::MultiScalarG2 ? rng_of({&S, &R, &T, &T, &T})
                   : mix                         ? rng_of({&R, &S, &S2, &T, &T, &T, &T})
                                                 : rng_of({&R, &S, &T, &T, &T});
    StatementProof sp = st.commit_and_prove(v, crs, rs);
    CHECK(com_x(sp, xvars) == xcoms && com_y(sp, yvars) == ycoms);
    CHECK(sp.equ_proofs.size() == st.size());
    CHECK(cat(sp.equ_proofs[0].pi) == pi && cat(sp.equ_proofs[0].theta) == th);
    CHECK(cat(sp.equ_proofs[2].pi) == pi && cat(sp.equ_proofs[2].theta) == th);
    ReplayRng r1 = rng_of({&T});
    EquProof single = other.prove(xvars, yvars, xcoms, ycoms, crs, r1);
    CHECK(cat(sp.equ_proofs[1].pi) == cat(single.pi) && cat(sp.equ_proofs[1].theta) == cat(single.theta));
    std::vector<bool> ok = st.verify(sp, crs);
    CHECK(ok.size() == st.size() && ok[0] && ok[2]);
    CHECK(ok[1] == other.verify(CProof{xcoms, ycoms, {sp.equ_proofs[1]}}, crs));
    if (mix) {
      ReplayRng r2 = rng_of({&T});
      EquProof p1 = e1.prove(as_g1v(xvars), ys2, sp.com_xg, sp.com_ys, crs, r2);
      CHECK(cat(sp.equ_proofs[3].pi) == cat(p1.pi) && cat(sp.equ_proofs[3].theta) == cat(p1.theta));
      CHECK(sp.equ_proofs[3].equ_type == EquType::MultiScalarG1);
      CHECK(ok[3] == e1.verify(CProof{sp.com_xg, sp.com_ys, {sp.equ_proofs[3]}}, crs));
    }
    bool threw = false;
    try {
      StatementProof bad = sp;
      bad.equ_proofs[1].pi.pop_back();
      st.verify(bad, crs);
    } catch (const Panic&) {
      threw = true;
    }
    CHECK(threw);
  }

  // shape asserts panic (prove.rs:106-113; verifier.rs:25-26)
  {
    bool threw = false;
    try {
      std::vector<A1> shortx(xvars.begin(), xvars.end() - 1);
      ReplayRng r = rng_of({&T});
      equ.prove(shortx, yvars, xcoms, ycoms, crs, r);
    } catch (const Panic&) {
      threw = true;
    }
    CHECK(threw);
    threw = false;
    try {
      CProof two = proof;
      two.equ_proofs.push_back(two.equ_proofs[0]);
      equ.verify(two, crs);
    } catch (const Panic&) {
      threw = true;
    }
    CHECK(threw);
    // lengths that come from the wire must panic BEFORE the C ABI reads them (the reference panics in pairing_sum /
    // left_mul, data_structures.rs:495,705): short pi, short theta, missing constants, ragged Gamma, short target
    auto panics = [&](const Equ& e, const CProof& p) {
      try {
        e.verify(p, crs);
      } catch (const Panic&) {
        return true;
      }
      return false;
    };
    {
      CProof p = proof;
      p.equ_proofs[0].pi.pop_back();
      CHECK(panics(equ, p));
      p = proof;
      p.equ_proofs[0].theta.clear();
      CHECK(panics(equ, p));
      p = proof;
      p.equ_proofs[0].pi[0].v.resize(p.equ_proofs[0].pi[0].v.size() / 2);
      CHECK(panics(equ, p));
      Equ e2 = equ;
      e2.a_consts.pop_back();
      CHECK(panics(e2, proof));
      e2 = equ;
      e2.b_consts.push_back(e2.b_consts[0]);
      CHECK(panics(e2, proof));
      e2 = equ;
      e2.gamma[0].pop_back();
      CHECK(panics(e2, proof));
      e2 = equ;
      e2.target.v.resize(e2.target.v.size() - 8);
      CHECK(panics(e2, proof));
      p = proof;
      p.xcoms.coms.clear();
      CHECK(panics(equ, p));
    }
    threw = false;
    try {
      Equ e2 = equ;
      e2.a_consts.pop_back();
      ReplayRng r = rng_of({&T});
      e2.prove(xvars, yvars, xcoms, ycoms, crs, r);
    } catch (const Panic&) {
      threw = true;
    }
    CHECK(threw);
  }
}

int main(int argc, char** argv) {
  if (argc < 2) {
    std::fprintf(stderr, "usage: %s case.bin\n", argv[0]);
    return 2;
  }
  std::ifstream f(argv[1], std::ios::binary);
  uint32_t hdr[4];
  f.read((char*)hdr, 16);
  Bytes s[24];
  for (auto& b : s) b = section(f);
  CRS crs({Com1{s[0]}, Com1{s[1]}}, {Com2{s[2]}, Com2{s[3]}}, G1Affine{s[4]}, G2Affine{s[5]}, GT{s[6]}, (int)hdr[0]);
  uint32_t ty = hdr[1], m = hdr[2], n = hdr[3];
#define ARGS crs, m, n, s[7], s[8], s[9], s[10], s[11], s[12], s[13], s[14], s[15], s[16], s[17], s[18], s[19], s[20], \
             s[21], s[22], s[23], (int)hdr[0]
  switch (ty) {
    case 0: run<G1Affine, G2Affine, GT, EquType::PairingProduct>(ARGS); break;
    case 1: run<G1Affine, Fr, G1Affine, EquType::MultiScalarG1>(ARGS); break;
    case 2: run<Fr, G2Affine, G2Affine, EquType::MultiScalarG2>(ARGS); break;
    case 3: run<Fr, Fr, Fr, EquType::Quadratic>(ARGS); break;
    default: return 2;
  }
  std::printf("OK %d\n", checks);
  return 0;
}
