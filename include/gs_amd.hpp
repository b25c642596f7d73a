// gs_amd.hpp -- C++17 host-side mirror of the reference's operator interface for the
// hot path, header-only on top of the C ABI (gs_amd.h).  Same names, argument
// meaning and error behaviour as the Rust API it stands in for (the reference's
// toolchain is absent from the build image, so the host layer a Rust shim would
// provide is written in C++; INTEGRATION.md shows the Rust-side binding):
//
//   CRS, CRS::generate_crs                       src/generator.rs:35-42, 81-118
//   Commit1 / Commit2 {coms, rand}, append        src/prover/commit.rs:18-56
//   commit_G1 / batch_commit_G1 / ..._scalar_to_B2 src/prover/commit.rs:59-256
//   EquType                                       src/statement.rs:42-49
//   PPE / MSMEG1 / MSMEG2 / QuadEqu               src/statement.rs:117-192
//     .commit_and_prove(xvars, yvars, crs, rng)   src/prover/prove.rs:29-52 (Provable)
//     .prove(xvars, yvars, xcoms, ycoms, crs, rng)
//     .verify(com_proof, crs) -> bool             src/verifier.rs:18-21   (Verifiable)
//   EquProof {pi, theta, equ_type, rand}, CProof  src/prover/prove.rs:55-69
//
// Values are byte strings in the boundary layout of gs_amd.h (arkworks' Montgomery
// limbs).  `Rng` is any type with `Fr fr()`; draws happen in the reference's order
// (R row-major, then S, then T).  Where the reference panics (assert_eq!,
// prove.rs:106-113, verifier.rs:25-26) these functions throw gs_amd::Panic.
#pragma once
#include <cstdint>
#include <cstring>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "gs_amd.h"

namespace gs_amd {

struct Panic : std::logic_error {
  using std::logic_error::logic_error;
};
inline void assert_eq(size_t a, size_t b, const char* what) {
  if (a != b) throw Panic(std::string("assertion failed: ") + what);
}

using Bytes = std::vector<uint8_t>;
struct Fr { Bytes v; };
struct G1Affine { Bytes v; };
struct G2Affine { Bytes v; };
struct GT { Bytes v; };  // PairingOutput
struct Com1 { Bytes v; };  // G1 || G1
struct Com2 { Bytes v; };  // G2 || G2
template <class T> using Matrix = std::vector<std::vector<T>>;
inline bool operator==(const Fr& a, const Fr& b) { return a.v == b.v; }
inline bool operator==(const Com1& a, const Com1& b) { return a.v == b.v; }
inline bool operator==(const Com2& a, const Com2& b) { return a.v == b.v; }

enum class EquType : uint8_t { PairingProduct = 0, MultiScalarG1 = 1, MultiScalarG2 = 2, Quadratic = 3 };

// one GPU context per CRS (gs_ctx_create + gs_set_crs), shared by copies of the CRS
struct Ctx {
  gs_ctx* c = nullptr;
  size_t sz[6] = {0, 0, 0, 0, 0, 0};  // Fq, Fr, G1, G2, GT, CRS
  int curve_id = 0;
  Ctx(int curve, int device) : curve_id(curve) {
    if (gs_sizes(curve, sz) != GS_OK) throw std::runtime_error("bad curve id");
    int rc = gs_ctx_create(curve, device, &c);
    if (rc != GS_OK) throw std::runtime_error("gs_ctx_create failed (no usable GPU; there is no CPU fallback)");
  }
  ~Ctx() { gs_ctx_destroy(c); }
  void chk(int rc) const {
    if (rc == GS_ERR_SHAPE) throw Panic(gs_last_error(c));
    if (rc != GS_OK) throw std::runtime_error(std::string("gs_amd error: ") + gs_last_error(c));
  }
};

// A caller's buffer page-locked for as long as the object lives (gs_host_register): batches handed over in it are
// moved by DMA directly instead of through the library's staging copy.
class PinnedRegion {
  const Ctx& ctx_;
  void* p_;

 public:
  PinnedRegion(const Ctx& ctx, void* p, size_t bytes) : ctx_(ctx), p_(p) { ctx_.chk(gs_host_register(ctx_.c, p, bytes)); }
  PinnedRegion(const PinnedRegion&) = delete;
  PinnedRegion& operator=(const PinnedRegion&) = delete;
  ~PinnedRegion() { gs_host_unregister(ctx_.c, p_); }
};

template <class T> inline Bytes cat(const std::vector<T>& xs) {
  Bytes o;
  for (const T& x : xs) o.insert(o.end(), x.v.begin(), x.v.end());
  return o;
}
inline Bytes cat(const Matrix<Fr>& m) {
  Bytes o;
  for (const auto& row : m)
    for (const Fr& x : row) o.insert(o.end(), x.v.begin(), x.v.end());
  return o;
}
template <class T> inline std::vector<T> split(const Bytes& b, size_t n) {
  std::vector<T> o(n);
  size_t w = n ? b.size() / n : 0;
  for (size_t i = 0; i < n; i++) o[i].v.assign(b.begin() + i * w, b.begin() + (i + 1) * w);
  return o;
}

struct CRS {  // generator.rs:35-42
  std::vector<Com1> u;
  std::vector<Com2> v;
  G1Affine g1_gen;
  G2Affine g2_gen;
  GT gt_gen;
  std::shared_ptr<Ctx> ctx;

  CRS(std::vector<Com1> u_, std::vector<Com2> v_, G1Affine g1, G2Affine g2, GT gt, int curve = GS_CURVE_BLS12_381,
      int device = 0)
      : u(std::move(u_)), v(std::move(v_)), g1_gen(std::move(g1)), g2_gen(std::move(g2)), gt_gen(std::move(gt)),
        ctx(std::make_shared<Ctx>(curve, device)) {
    Bytes flat = cat(u);
    Bytes t = cat(v);
    flat.insert(flat.end(), t.begin(), t.end());
    flat.insert(flat.end(), g1_gen.v.begin(), g1_gen.v.end());
    flat.insert(flat.end(), g2_gen.v.begin(), g2_gen.v.end());
    flat.insert(flat.end(), gt_gen.v.begin(), gt_gen.v.end());
    assert_eq(flat.size(), ctx->sz[5], "CRS size");
    ctx->chk(gs_set_crs(ctx->c, flat.data()));
  }
  // AbstractCrs::generate_crs (generator.rs:81-118); the generators come from the caller, the
  // scalars a1, a2, t1, t2 from the rng in the reference's order (:90-93)
  template <class Rng>
  static CRS generate_crs(const G1Affine& p1, const G2Affine& p2, Rng& rng, int curve = GS_CURVE_BLS12_381,
                          int device = 0) {
    Ctx tmp(curve, device);
    Bytes sc;
    for (int i = 0; i < 4; i++) {
      Fr s = rng.fr();
      sc.insert(sc.end(), s.v.begin(), s.v.end());
    }
    Bytes raw(tmp.sz[5]);
    tmp.chk(gs_crs_generate(tmp.c, p1.v.data(), p2.v.data(), sc.data(), raw.data()));
    size_t g1 = tmp.sz[2], g2 = tmp.sz[3], o = 0;
    auto take = [&](size_t n) {
      Bytes b(raw.begin() + o, raw.begin() + o + n);
      o += n;
      return b;
    };
    std::vector<Com1> u{{take(2 * g1)}, {take(2 * g1)}};
    std::vector<Com2> v{{take(2 * g2)}, {take(2 * g2)}};
    G1Affine a{take(g1)};
    G2Affine b{take(g2)};
    GT t{take(tmp.sz[4])};
    return CRS(u, v, a, b, t, curve, device);
  }
};

template <class C> struct CommitT {  // commit.rs:18-28
  std::vector<C> coms;
  Matrix<Fr> rand;
  bool operator==(const CommitT& o) const {
    if (coms.size() != o.coms.size() || rand.size() != o.rand.size()) return false;
    for (size_t i = 0; i < coms.size(); i++)
      if (!(coms[i] == o.coms[i])) return false;
    for (size_t i = 0; i < rand.size(); i++) {
      if (rand[i].size() != o.rand[i].size()) return false;
      for (size_t j = 0; j < rand[i].size(); j++)
        if (!(rand[i][j] == o.rand[i][j])) return false;
    }
    return true;
  }
  void append(CommitT& other) {  // commit.rs:43-51
    assert_eq(coms.size(), rand.size(), "self.coms.len() == self.rand.len()");
    assert_eq(other.coms.size(), other.rand.size(), "other.coms.len() == other.rand.len()");
    coms.insert(coms.end(), other.coms.begin(), other.coms.end());
    rand.insert(rand.end(), other.rand.begin(), other.rand.end());
    other.coms.clear();
    other.rand.clear();
  }
};
using Commit1 = CommitT<Com1>;
using Commit2 = CommitT<Com2>;

namespace detail {
template <class Rng> Matrix<Fr> draw(Rng& rng, size_t rows, size_t cols) {
  Matrix<Fr> m(rows);
  for (auto& r : m)
    for (size_t j = 0; j < cols; j++) r.push_back(rng.fr());
  return m;
}
template <class Com, class V, class Rng, class F>
CommitT<Com> batch_commit(const std::vector<V>& vars, const CRS& key, Rng& rng, size_t cols, size_t out_sz, F fn) {
  CommitT<Com> c;
  c.rand = draw(rng, vars.size(), cols);
  if (vars.empty()) return c;
  Bytes in = cat(vars), r = cat(c.rand), out(vars.size() * out_sz);
  key.ctx->chk(fn(key.ctx->c, vars.size(), in.data(), r.data(), out.data()));
  c.coms = split<Com>(out, vars.size());
  return c;
}
}  // namespace detail

// commit.rs:78-100, 178-200, 125-156, 225-256 (+ the single-element forms :59-75, 103-122, 159-175, 203-222)
template <class Rng> Commit1 batch_commit_G1(const std::vector<G1Affine>& xvars, const CRS& key, Rng& rng) {
  return detail::batch_commit<Com1>(xvars, key, rng, 2, 2 * key.ctx->sz[2], gs_commit_g1);
}
template <class Rng> Commit2 batch_commit_G2(const std::vector<G2Affine>& yvars, const CRS& key, Rng& rng) {
  return detail::batch_commit<Com2>(yvars, key, rng, 2, 2 * key.ctx->sz[3], gs_commit_g2);
}
template <class Rng> Commit1 batch_commit_scalar_to_B1(const std::vector<Fr>& xs, const CRS& key, Rng& rng) {
  return detail::batch_commit<Com1>(xs, key, rng, 1, 2 * key.ctx->sz[2], gs_commit_fr_b1);
}
template <class Rng> Commit2 batch_commit_scalar_to_B2(const std::vector<Fr>& ys, const CRS& key, Rng& rng) {
  return detail::batch_commit<Com2>(ys, key, rng, 1, 2 * key.ctx->sz[3], gs_commit_fr_b2);
}
template <class Rng> Commit1 commit_G1(const G1Affine& x, const CRS& key, Rng& rng) {
  return batch_commit_G1(std::vector<G1Affine>{x}, key, rng);
}
template <class Rng> Commit2 commit_G2(const G2Affine& y, const CRS& key, Rng& rng) {
  return batch_commit_G2(std::vector<G2Affine>{y}, key, rng);
}
template <class Rng> Commit1 commit_scalar_to_B1(const Fr& x, const CRS& key, Rng& rng) {
  return batch_commit_scalar_to_B1(std::vector<Fr>{x}, key, rng);
}
template <class Rng> Commit2 commit_scalar_to_B2(const Fr& y, const CRS& key, Rng& rng) {
  return batch_commit_scalar_to_B2(std::vector<Fr>{y}, key, rng);
}
// overload selection by witness type (used by commit_and_prove)
template <class Rng> Commit1 batch_commit_x(const std::vector<G1Affine>& x, const CRS& k, Rng& r) { return batch_commit_G1(x, k, r); }
template <class Rng> Commit1 batch_commit_x(const std::vector<Fr>& x, const CRS& k, Rng& r) { return batch_commit_scalar_to_B1(x, k, r); }
template <class Rng> Commit2 batch_commit_y(const std::vector<G2Affine>& y, const CRS& k, Rng& r) { return batch_commit_G2(y, k, r); }
template <class Rng> Commit2 batch_commit_y(const std::vector<Fr>& y, const CRS& k, Rng& r) { return batch_commit_scalar_to_B2(y, k, r); }

struct EquProof {  // prove.rs:55-61
  std::vector<Com2> pi;
  std::vector<Com1> theta;
  EquType equ_type;
  Matrix<Fr> rand;
};
struct CProof {  // prove.rs:64-69
  Commit1 xcoms;
  Commit2 ycoms;
  std::vector<EquProof> equ_proofs;
};

template <class T> struct is_scalar { static constexpr bool value = false; };
template <> struct is_scalar<Fr> { static constexpr bool value = true; };

// statement.rs:117-192: a_consts pair with the Y variables and live in A1, b_consts in A2
template <class A1, class A2, class AT, EquType TYPE> struct Equation {
  std::vector<A1> a_consts;
  std::vector<A2> b_consts;
  Matrix<Fr> gamma;
  AT target;
  static constexpr size_t KX = is_scalar<A1>::value ? 1 : 2;  // columns of R = number of pi elements
  static constexpr size_t KY = is_scalar<A2>::value ? 1 : 2;  // columns of S = number of theta elements

  EquType get_type() const { return TYPE; }

  // a_consts pair with the n Y variables, b_consts with the m X variables, Gamma is m x n
  void check_statement_shape(size_t m, size_t n) const {
    assert_eq(a_consts.size(), n, "a_consts.len() == yvars.len()");
    assert_eq(b_consts.size(), m, "b_consts.len() == xvars.len()");
    assert_eq(gamma.size(), m, "gamma.len() == xvars.len()");
    for (const auto& row : gamma) assert_eq(row.size(), n, "gamma[i].len() == yvars.len()");
  }

  template <class Rng>
  CProof commit_and_prove(const std::vector<A1>& xvars, const std::vector<A2>& yvars, const CRS& crs, Rng& rng) const {
    Commit1 xcoms = batch_commit_x(xvars, crs, rng);
    Commit2 ycoms = batch_commit_y(yvars, crs, rng);
    CProof p{xcoms, ycoms, {}};
    p.equ_proofs.push_back(prove(xvars, yvars, xcoms, ycoms, crs, rng));
    return p;
  }

  template <class Rng>
  EquProof prove(const std::vector<A1>& xvars, const std::vector<A2>& yvars, const Commit1& xcoms,
                 const Commit2& ycoms, const CRS& crs, Rng& rng) const {
    // the reference's shape asserts (prove.rs:106-113); empty lists panic there on rand[0]
    assert_eq(xvars.size(), xcoms.rand.size(), "xvars.len() == xcoms.rand.len()");
    assert_eq(gamma.size(), xcoms.rand.size(), "gamma.len() == xcoms.rand.len()");
    if (xcoms.rand.empty() || ycoms.rand.empty() || gamma.empty()) throw Panic("index out of bounds: rand[0]");
    assert_eq(xcoms.rand[0].size(), KX, "xcoms.rand[0].len()");
    assert_eq(yvars.size(), ycoms.rand.size(), "yvars.len() == ycoms.rand.len()");
    assert_eq(gamma[0].size(), ycoms.rand.size(), "gamma[0].len() == ycoms.rand.len()");
    assert_eq(ycoms.rand[0].size(), KY, "ycoms.rand[0].len()");
    size_t m = xvars.size(), n = yvars.size();
    // what the reference's left_mul / pairing_sum would panic on later (data_structures.rs:495,705): every row of
    // Gamma has n entries, one constant per variable of the other side, uniform randomness rows
    check_statement_shape(m, n);
    for (const auto& r : xcoms.rand) assert_eq(r.size(), KX, "xcoms.rand[i].len()");
    for (const auto& r : ycoms.rand) assert_eq(r.size(), KY, "ycoms.rand[j].len()");
    Matrix<Fr> T = detail::draw(rng, KY, KX);
    Bytes X = cat(xvars), Y = cat(yvars), A = cat(a_consts), B = cat(b_consts), G = cat(gamma), R = cat(xcoms.rand),
          S = cat(ycoms.rand), Tb = cat(T);
    const Ctx& cx = *crs.ctx;
    const size_t sx = is_scalar<A1>::value ? cx.sz[1] : cx.sz[2], sy = is_scalar<A2>::value ? cx.sz[1] : cx.sz[3];
    assert_eq(X.size(), m * sx, "xvars bytes");
    assert_eq(Y.size(), n * sy, "yvars bytes");
    assert_eq(A.size(), n * sx, "a_consts bytes");
    assert_eq(B.size(), m * sy, "b_consts bytes");
    assert_eq(G.size(), m * n * cx.sz[1], "gamma bytes");
    assert_eq(R.size(), m * KX * cx.sz[1], "xcoms.rand bytes");
    assert_eq(S.size(), n * KY * cx.sz[1], "ycoms.rand bytes");
    Bytes pi(KX * 2 * cx.sz[3]), th(KY * 2 * cx.sz[2]);
    cx.chk(gs_prove_batch(cx.c, (int)TYPE, 1, (int)m, (int)n, X.data(), Y.data(), A.data(), B.data(), G.data(),
                          R.data(), S.data(), Tb.data(), nullptr, nullptr, pi.data(), th.data()));
    EquProof pf{split<Com2>(pi, KX), split<Com1>(th, KY), TYPE, T};
    assert_eq(pf.pi.size(), KX, "pi.len()");
    assert_eq(pf.theta.size(), KY, "theta.len()");
    return pf;
  }

  bool verify(const CProof& com_proof, const CRS& crs) const {  // verifier.rs:23-157
    assert_eq(com_proof.equ_proofs.size(), 1, "com_proof.equ_proofs.len() == 1");
    if (com_proof.equ_proofs[0].equ_type != TYPE) throw Panic("assertion failed: equation type matches the proof's");
    const EquProof& pf = com_proof.equ_proofs[0];
    size_t m = com_proof.xcoms.coms.size(), n = com_proof.ycoms.coms.size();
    // Lengths come straight from the wire (Vec<_> prefixes): everything the C ABI will index is checked HERE, where
    // the reference panics in pairing_sum / left_mul (data_structures.rs:495,705) -- never read past a short buffer.
    if (m == 0 || n == 0) throw Panic("index out of bounds: empty commitment list");
    check_statement_shape(m, n);
    assert_eq(pf.pi.size(), KX, "pi.len()");
    assert_eq(pf.theta.size(), KY, "theta.len()");
    const Ctx& cxs = *crs.ctx;
    const size_t tsz = TYPE == EquType::PairingProduct ? cxs.sz[4] : TYPE == EquType::MultiScalarG1 ? cxs.sz[2]
                       : TYPE == EquType::MultiScalarG2 ? cxs.sz[3] : cxs.sz[1];
    assert_eq(target.v.size(), tsz, "target size");
    Bytes A = cat(a_consts), B = cat(b_consts), G = cat(gamma), xc = cat(com_proof.xcoms.coms),
          yc = cat(com_proof.ycoms.coms), pi = cat(pf.pi), th = cat(pf.theta);
    uint8_t ok = 0;
    const Ctx& cx = *crs.ctx;
    const size_t sx = is_scalar<A1>::value ? cx.sz[1] : cx.sz[2], sy = is_scalar<A2>::value ? cx.sz[1] : cx.sz[3];
    assert_eq(A.size(), n * sx, "a_consts bytes");
    assert_eq(B.size(), m * sy, "b_consts bytes");
    assert_eq(G.size(), m * n * cx.sz[1], "gamma bytes");
    assert_eq(xc.size(), m * 2 * cx.sz[2], "xcoms bytes");
    assert_eq(yc.size(), n * 2 * cx.sz[3], "ycoms bytes");
    assert_eq(pi.size(), KX * 2 * cx.sz[3], "pi bytes");
    assert_eq(th.size(), KY * 2 * cx.sz[2], "theta bytes");
    cx.chk(gs_verify_batch(cx.c, (int)TYPE, 1, (int)m, (int)n, A.data(), B.data(), G.data(), target.v.data(),
                           xc.data(), yc.data(), pi.data(), th.data(), &ok));
    return ok == 1;
  }
};

using PPE = Equation<G1Affine, G2Affine, GT, EquType::PairingProduct>;
using MSMEG1 = Equation<G1Affine, Fr, G1Affine, EquType::MultiScalarG1>;
using MSMEG2 = Equation<Fr, G2Affine, G2Affine, EquType::MultiScalarG2>;
using QuadEqu = Equation<Fr, Fr, Fr, EquType::Quadratic>;

// ---- `pub type Statement = Vec<dyn Equ>` (statement.rs:24-28,109): "a list of equations ... defined with respect to
// the list of variables that span across ALL equations" -- unusable in the reference (a Vec of an unsized type), made
// usable here.  The variables are four lists -- G1 variables xg, G2 variables yg, scalar variables xs (committed into
// B1) and ys (into B2); a PPE is over (xg, yg), an MSMEG1 over (xg, ys), an MSMEG2 over (xs, yg), a QuadEqu over
// (xs, ys) (statement.rs:117-192) -- each committed ONCE, and every equation of every type gets its own EquProof
// against those commitments in ONE call of the engine (gs_prove_mixed / gs_verify_mixed with shared_vars parts: the
// launches of the parts are merged on the device).  Draw order: the commit randomness of xg, yg, xs, ys, then T of
// equation 0, 1, ...  Exactly what batch_commit_* followed by Provable::prove per equation with shared Commit1 /
// Commit2 does in the reference.
struct StatementVars {
  std::vector<G1Affine> xg;
  std::vector<G2Affine> yg;
  std::vector<Fr> xs, ys;
};
struct StatementProof {
  Commit1 com_xg, com_xs;
  Commit2 com_yg, com_ys;
  std::vector<EquProof> equ_proofs;  // one per equation, in the statement's order
};
class Statement {
  struct Item {  // one equation, type-erased to bytes
    EquType ty;
    size_t na, nb, rows, cols;  // a_consts, b_consts, Gamma rows / columns (checked against the groups' sizes)
    Bytes A, B, G, target;
    bool ragged;
  };
  std::vector<Item> items;
  static bool xgroup(EquType t) { return t == EquType::PairingProduct || t == EquType::MultiScalarG1; }
  static bool ygroup(EquType t) { return t == EquType::PairingProduct || t == EquType::MultiScalarG2; }
  struct Part {
    EquType ty;
    std::vector<size_t> idx;
    Bytes A, B, G, T, target, pi, theta;
    std::vector<uint8_t> ok;
  };
  std::vector<Part> parts() const {
    std::vector<Part> ps;
    for (size_t i = 0; i < items.size(); i++) {
      size_t k = 0;
      while (k < ps.size() && ps[k].ty != items[i].ty) k++;
      if (k == ps.size()) ps.push_back(Part{items[i].ty, {}, {}, {}, {}, {}, {}, {}, {}, {}});
      ps[k].idx.push_back(i);
    }
    return ps;
  }
  void check(const Item& it, size_t m, size_t n) const {
    assert_eq(it.na, n, "a_consts.len() == yvars.len()");
    assert_eq(it.nb, m, "b_consts.len() == xvars.len()");
    assert_eq(it.rows, m, "gamma.len() == xvars.len()");
    if (it.ragged || it.cols != n) throw Panic("assertion failed: gamma[i].len() == yvars.len()");
  }

 public:
  size_t size() const { return items.size(); }
  template <class A1, class A2, class AT, EquType TYPE> void push(const Equation<A1, A2, AT, TYPE>& e) {
    Item it{TYPE, e.a_consts.size(), e.b_consts.size(), e.gamma.size(), e.gamma.empty() ? 0 : e.gamma[0].size(),
            cat(e.a_consts), cat(e.b_consts), cat(e.gamma), e.target.v, false};
    for (const auto& row : e.gamma) it.ragged = it.ragged || row.size() != it.cols;
    items.push_back(std::move(it));
  }

  template <class Rng> StatementProof commit_and_prove(const StatementVars& v, const CRS& crs, Rng& rng) const {
    StatementProof out;
    out.com_xg = batch_commit_G1(v.xg, crs, rng);
    out.com_yg = batch_commit_G2(v.yg, crs, rng);
    out.com_xs = batch_commit_scalar_to_B1(v.xs, crs, rng);
    out.com_ys = batch_commit_scalar_to_B2(v.ys, crs, rng);
    std::vector<Matrix<Fr>> Ts;
    for (const Item& it : items) Ts.push_back(detail::draw(rng, ygroup(it.ty) ? 2 : 1, xgroup(it.ty) ? 2 : 1));
    const Ctx& cx = *crs.ctx;
    std::vector<Part> ps = parts();
    if (ps.size() > GS_MIXED_MAX) throw Panic("more equation types than a mixed call takes");
    std::vector<gs_prove_part> cp(ps.size());
    Bytes Xg = cat(v.xg), Yg = cat(v.yg), Xs = cat(v.xs), Ys = cat(v.ys);
    Bytes Rg = cat(out.com_xg.rand), Sg = cat(out.com_yg.rand), Rs = cat(out.com_xs.rand), Ss = cat(out.com_ys.rand);
    for (size_t k = 0; k < ps.size(); k++) {
      Part& p = ps[k];
      const bool xg = xgroup(p.ty), yg = ygroup(p.ty);
      const size_t m = xg ? v.xg.size() : v.xs.size(), n = yg ? v.yg.size() : v.ys.size(), kx = xg ? 2 : 1, ky = yg ? 2 : 1;
      if (m == 0 || n == 0) throw Panic("index out of bounds: rand[0]");
      for (size_t i : p.idx) {
        check(items[i], m, n);
        auto app = [](Bytes& d, const Bytes& s) { d.insert(d.end(), s.begin(), s.end()); };
        app(p.A, items[i].A);
        app(p.B, items[i].B);
        app(p.G, items[i].G);
        app(p.T, cat(Ts[i]));
      }
      const size_t E = p.idx.size(), sx = xg ? cx.sz[2] : cx.sz[1], sy = yg ? cx.sz[3] : cx.sz[1];
      assert_eq(p.A.size(), E * n * sx, "a_consts bytes");
      assert_eq(p.B.size(), E * m * sy, "b_consts bytes");
      assert_eq(p.G.size(), E * m * n * cx.sz[1], "gamma bytes");
      p.pi.assign(E * kx * 2 * cx.sz[3], 0);
      p.theta.assign(E * ky * 2 * cx.sz[2], 0);
      cp[k] = gs_prove_part{(int)p.ty, E, (int)m, (int)n, (xg ? Xg : Xs).data(), (yg ? Yg : Ys).data(), p.A.data(),
                            p.B.data(), p.G.data(), (xg ? Rg : Rs).data(), (yg ? Sg : Ss).data(), p.T.data(), nullptr,
                            nullptr, p.pi.data(), p.theta.data(), 1};
    }
    cx.chk(gs_prove_mixed(cx.c, (int)cp.size(), cp.data()));
    out.equ_proofs.resize(items.size());
    for (const Part& p : ps) {
      const size_t kx = xgroup(p.ty) ? 2 : 1, ky = ygroup(p.ty) ? 2 : 1;
      auto pis = split<Com2>(p.pi, p.idx.size() * kx);
      auto ths = split<Com1>(p.theta, p.idx.size() * ky);
      for (size_t e = 0; e < p.idx.size(); e++)
        out.equ_proofs[p.idx[e]] = EquProof{{pis.begin() + e * kx, pis.begin() + (e + 1) * kx},
                                            {ths.begin() + e * ky, ths.begin() + (e + 1) * ky}, p.ty, Ts[p.idx[e]]};
    }
    return out;
  }

  // one verdict per equation, in the statement's order; the statement holds iff all are true
  std::vector<bool> verify(const StatementProof& pr, const CRS& crs) const {
    assert_eq(pr.equ_proofs.size(), items.size(), "proof.equ_proofs.len() == statement.len()");
    const Ctx& cx = *crs.ctx;
    std::vector<Part> ps = parts();
    if (ps.size() > GS_MIXED_MAX) throw Panic("more equation types than a mixed call takes");
    std::vector<gs_verify_part> cp(ps.size());
    Bytes Cxg = cat(pr.com_xg.coms), Cyg = cat(pr.com_yg.coms), Cxs = cat(pr.com_xs.coms), Cys = cat(pr.com_ys.coms);
    for (size_t k = 0; k < ps.size(); k++) {
      Part& p = ps[k];
      const bool xg = xgroup(p.ty), yg = ygroup(p.ty);
      const size_t m = (xg ? pr.com_xg : pr.com_xs).coms.size(), n = (yg ? pr.com_yg : pr.com_ys).coms.size();
      const size_t kx = xg ? 2 : 1, ky = yg ? 2 : 1;
      if (m == 0 || n == 0) throw Panic("index out of bounds: empty commitment list");
      const size_t tsz = p.ty == EquType::PairingProduct ? cx.sz[4] : p.ty == EquType::MultiScalarG1 ? cx.sz[2]
                         : p.ty == EquType::MultiScalarG2 ? cx.sz[3] : cx.sz[1];
      auto app = [](Bytes& d, const Bytes& s) { d.insert(d.end(), s.begin(), s.end()); };
      for (size_t i : p.idx) {
        check(items[i], m, n);
        const EquProof& pf = pr.equ_proofs[i];
        if (pf.equ_type != p.ty) throw Panic("assertion failed: equation type matches the proof's");
        assert_eq(pf.pi.size(), kx, "pi.len()");
        assert_eq(pf.theta.size(), ky, "theta.len()");
        assert_eq(items[i].target.size(), tsz, "target size");
        app(p.A, items[i].A);
        app(p.B, items[i].B);
        app(p.G, items[i].G);
        app(p.target, items[i].target);
        app(p.pi, cat(pf.pi));
        app(p.theta, cat(pf.theta));
      }
      const size_t E = p.idx.size(), sx = xg ? cx.sz[2] : cx.sz[1], sy = yg ? cx.sz[3] : cx.sz[1];
      assert_eq(p.A.size(), E * n * sx, "a_consts bytes");
      assert_eq(p.B.size(), E * m * sy, "b_consts bytes");
      assert_eq(p.G.size(), E * m * n * cx.sz[1], "gamma bytes");
      assert_eq(p.pi.size(), E * kx * 2 * cx.sz[3], "pi bytes");
      assert_eq(p.theta.size(), E * ky * 2 * cx.sz[2], "theta bytes");
      assert_eq((xg ? Cxg : Cxs).size(), m * 2 * cx.sz[2], "xcoms bytes");
      assert_eq((yg ? Cyg : Cys).size(), n * 2 * cx.sz[3], "ycoms bytes");
      p.ok.assign(E, 0);
      cp[k] = gs_verify_part{(int)p.ty, E, (int)m, (int)n, p.A.data(), p.B.data(), p.G.data(), p.target.data(),
                             (xg ? Cxg : Cxs).data(), (yg ? Cyg : Cys).data(), p.pi.data(), p.theta.data(), p.ok.data(), 1};
    }
    cx.chk(gs_verify_mixed(cx.c, (int)cp.size(), cp.data()));
    std::vector<bool> out(items.size(), false);
    for (const Part& p : ps)
      for (size_t e = 0; e < p.idx.size(); e++) out[p.idx[e]] = p.ok[e] != 0;
    return out;
  }
};

// ---- canonical wire format (ark-serialize; the derives at data_structures.rs:128,132, commit.rs:18,24,
// prove.rs:55, statement.rs:117-179, generator.rs:35).  Framing here, element codecs in the library
// (gs_wire_*).  serialize_compressed / serialize_uncompressed / deserialize_compressed<T> /
// deserialize_uncompressed<T> follow ark-serialize's names; deserialisation validates like Validate::Yes
// and throws SerializationError where arkworks returns SerializationError::InvalidData.
struct SerializationError : std::runtime_error {
  using std::runtime_error::runtime_error;
};

namespace wire {
enum Kind { K_G1 = 0, K_G2 = 1, K_FR = 2, K_GT = 3, K_RAW = 4 };
inline size_t elem_size(const Ctx& cx, Kind k) {  // boundary bytes
  return k == K_G1 ? cx.sz[2] : k == K_G2 ? cx.sz[3] : k == K_FR ? cx.sz[1] : cx.sz[4];
}
inline size_t wire_size(const Ctx& cx, Kind k, bool compressed) {
  size_t w[6];
  gs_wire_sizes(cx.curve_id, w);
  return k == K_G1 ? w[compressed ? 0 : 1] : k == K_G2 ? w[compressed ? 2 : 3] : k == K_FR ? w[4] : w[5];
}

struct Writer {
  struct Leaf {
    Kind k;
    Bytes v;
  };
  std::vector<Leaf> leaves;
  void raw(Bytes b) { leaves.push_back({K_RAW, std::move(b)}); }
  void len(size_t n) {
    Bytes b(8);
    for (int i = 0; i < 8; i++) b[i] = (uint8_t)((uint64_t)n >> (8 * i));
    raw(b);
  }
  void put(const G1Affine& x) { leaves.push_back({K_G1, x.v}); }
  void put(const G2Affine& x) { leaves.push_back({K_G2, x.v}); }
  void put(const Fr& x) { leaves.push_back({K_FR, x.v}); }
  void put(const GT& x) { leaves.push_back({K_GT, x.v}); }
  void put(const Com1& c) {
    size_t h = c.v.size() / 2;
    leaves.push_back({K_G1, Bytes(c.v.begin(), c.v.begin() + h)});
    leaves.push_back({K_G1, Bytes(c.v.begin() + h, c.v.end())});
  }
  void put(const Com2& c) {
    size_t h = c.v.size() / 2;
    leaves.push_back({K_G2, Bytes(c.v.begin(), c.v.begin() + h)});
    leaves.push_back({K_G2, Bytes(c.v.begin() + h, c.v.end())});
  }
  template <class T> void vec(const std::vector<T>& xs) {
    len(xs.size());
    for (const T& x : xs) put(x);
  }
  void mat(const Matrix<Fr>& m) {
    len(m.size());
    for (const auto& row : m) vec(row);
  }
  Bytes run(const Ctx& cx, bool compressed) const {
    Bytes enc[4];
    size_t pos[4] = {0, 0, 0, 0};
    for (int k = 0; k < 4; k++) {
      Bytes in;
      size_t n = 0;
      for (const Leaf& l : leaves)
        if (l.k == k) {
          in.insert(in.end(), l.v.begin(), l.v.end());
          n++;
        }
      if (!n) continue;
      enc[k].resize(n * wire_size(cx, (Kind)k, compressed));
      int rc = k == K_G1   ? gs_wire_encode_g1(cx.c, n, compressed, in.data(), enc[k].data())
               : k == K_G2 ? gs_wire_encode_g2(cx.c, n, compressed, in.data(), enc[k].data())
               : k == K_FR ? gs_wire_encode_fr(cx.c, n, in.data(), enc[k].data())
                           : gs_wire_encode_gt(cx.c, n, in.data(), enc[k].data());
      cx.chk(rc);
    }
    Bytes out;
    for (const Leaf& l : leaves) {
      if (l.k == K_RAW) {
        out.insert(out.end(), l.v.begin(), l.v.end());
      } else {
        size_t w = wire_size(cx, l.k, compressed);
        out.insert(out.end(), enc[l.k].begin() + pos[l.k], enc[l.k].begin() + pos[l.k] + w);
        pos[l.k] += w;
      }
    }
    return out;
  }
};

struct Reader {
  struct H {
    Kind k;
    size_t i;
  };
  const Bytes& b;
  const Ctx& cx;
  bool compressed;
  size_t o = 0;
  Bytes req[4], val[4];
  size_t cnt[4] = {0, 0, 0, 0};
  Reader(const Bytes& data, const Ctx& c, bool comp) : b(data), cx(c), compressed(comp) {}
  const uint8_t* take(size_t n) {
    if (o + n > b.size()) throw SerializationError("InvalidData: truncated input");
    const uint8_t* p = b.data() + o;
    o += n;
    return p;
  }
  size_t len() {
    const uint8_t* p = take(8);
    uint64_t n = 0;
    for (int i = 0; i < 8; i++) n |= (uint64_t)p[i] << (8 * i);
    if (n > b.size()) throw SerializationError("InvalidData: length prefix exceeds the input");
    return (size_t)n;
  }
  uint8_t u8() { return *take(1); }
  H elem(Kind k) {
    size_t w = wire_size(cx, k, compressed);
    const uint8_t* p = take(w);
    req[k].insert(req[k].end(), p, p + w);
    return {k, cnt[k]++};
  }
  std::vector<H> vec(Kind k, size_t per = 1) {
    size_t n = len();
    std::vector<H> hs;
    for (size_t i = 0; i < n * per; i++) hs.push_back(elem(k));
    return hs;
  }
  Matrix<H> mat() {
    size_t n = len();
    Matrix<H> m(n);
    for (auto& row : m) row = vec(K_FR);
    return m;
  }
  void finish(bool validate = true) {
    if (o != b.size()) throw SerializationError("InvalidData: trailing bytes");
    for (int k = 0; k < 4; k++) {
      if (!cnt[k]) continue;
      val[k].resize(cnt[k] * elem_size(cx, (Kind)k));
      Bytes ok(cnt[k]);
      int rc = k == K_G1   ? gs_wire_decode_g1(cx.c, cnt[k], compressed, validate, req[k].data(), val[k].data(), ok.data())
               : k == K_G2 ? gs_wire_decode_g2(cx.c, cnt[k], compressed, validate, req[k].data(), val[k].data(), ok.data())
               : k == K_FR ? gs_wire_decode_fr(cx.c, cnt[k], req[k].data(), val[k].data(), ok.data())
                           : gs_wire_decode_gt(cx.c, cnt[k], validate, req[k].data(), val[k].data(), ok.data());
      cx.chk(rc);
      for (uint8_t f : ok)
        if (!f) throw SerializationError("InvalidData: element rejected");
    }
  }
  Bytes get(H h) const {
    size_t e = elem_size(cx, h.k);
    return Bytes(val[h.k].begin() + h.i * e, val[h.k].begin() + (h.i + 1) * e);
  }
  template <class Com> std::vector<Com> coms(const std::vector<H>& hs) const {
    std::vector<Com> out;
    for (size_t i = 0; i + 1 < hs.size(); i += 2) {
      Bytes a = get(hs[i]), c = get(hs[i + 1]);
      a.insert(a.end(), c.begin(), c.end());
      out.push_back(Com{a});
    }
    return out;
  }
  template <class T> std::vector<T> vals(const std::vector<H>& hs) const {
    std::vector<T> out;
    for (const H& h : hs) out.push_back(T{get(h)});
    return out;
  }
  Matrix<Fr> frs(const Matrix<H>& m) const {
    Matrix<Fr> out;
    for (const auto& row : m) out.push_back(vals<Fr>(row));
    return out;
  }
};
template <class T> struct KindOf;
template <> struct KindOf<G1Affine> { static constexpr Kind K = K_G1; };
template <> struct KindOf<G2Affine> { static constexpr Kind K = K_G2; };
template <> struct KindOf<Fr> { static constexpr Kind K = K_FR; };
template <> struct KindOf<GT> { static constexpr Kind K = K_GT; };

inline void describe(Writer& w, const Commit1& c) { w.vec(c.coms); w.mat(c.rand); }
inline void describe(Writer& w, const Commit2& c) { w.vec(c.coms); w.mat(c.rand); }
inline void describe(Writer& w, const EquProof& p) {
  w.vec(p.pi);
  w.vec(p.theta);
  w.raw(Bytes{(uint8_t)p.equ_type});
  w.mat(p.rand);
}
inline void describe(Writer& w, const CRS& c) {
  w.vec(c.u);
  w.vec(c.v);
  w.put(c.g1_gen);
  w.put(c.g2_gen);
  w.put(c.gt_gen);
}
template <class A1, class A2, class AT, EquType TY> void describe(Writer& w, const Equation<A1, A2, AT, TY>& e) {
  w.vec(e.a_consts);
  w.vec(e.b_consts);
  w.mat(e.gamma);
  w.put(e.target);
}

template <class T> struct Parse;
template <> struct Parse<Commit1> {
  static Commit1 run(Reader& r) {
    auto c = r.vec(K_G1, 2);
    auto m = r.mat();
    r.finish();
    return Commit1{r.coms<Com1>(c), r.frs(m)};
  }
};
template <> struct Parse<Commit2> {
  static Commit2 run(Reader& r) {
    auto c = r.vec(K_G2, 2);
    auto m = r.mat();
    r.finish();
    return Commit2{r.coms<Com2>(c), r.frs(m)};
  }
};
template <> struct Parse<EquProof> {
  static EquProof run(Reader& r) {
    auto pi = r.vec(K_G2, 2);
    auto th = r.vec(K_G1, 2);
    uint8_t ty = r.u8();
    if (ty > 3) throw SerializationError("InvalidData: EquType");
    auto m = r.mat();
    r.finish();
    return EquProof{r.coms<Com2>(pi), r.coms<Com1>(th), (EquType)ty, r.frs(m)};
  }
};
template <class A1, class A2, class AT, EquType TY> struct Parse<Equation<A1, A2, AT, TY>> {
  static Equation<A1, A2, AT, TY> run(Reader& r) {
    auto a = r.vec(KindOf<A1>::K);
    auto b = r.vec(KindOf<A2>::K);
    auto g = r.mat();
    auto t = r.elem(KindOf<AT>::K);
    r.finish();
    Equation<A1, A2, AT, TY> e;
    e.a_consts = r.vals<A1>(a);
    e.b_consts = r.vals<A2>(b);
    e.gamma = r.frs(g);
    e.target = AT{r.get(t)};
    return e;
  }
};
}  // namespace wire

template <class T> Bytes serialize_compressed(const T& x, const CRS& crs) {
  wire::Writer w;
  wire::describe(w, x);
  return w.run(*crs.ctx, true);
}
template <class T> Bytes serialize_uncompressed(const T& x, const CRS& crs) {
  wire::Writer w;
  wire::describe(w, x);
  return w.run(*crs.ctx, false);
}
template <class T> T deserialize_compressed(const Bytes& b, const CRS& crs) {
  wire::Reader r(b, *crs.ctx, true);
  return wire::Parse<T>::run(r);
}
template <class T> T deserialize_uncompressed(const Bytes& b, const CRS& crs) {
  wire::Reader r(b, *crs.ctx, false);
  return wire::Parse<T>::run(r);
}
// the CRS itself has no context to borrow: it is decoded on a fresh one (curve, device), then installed
inline CRS deserialize_crs(const Bytes& b, bool compressed, int curve = GS_CURVE_BLS12_381, int device = 0) {
  Ctx tmp(curve, device);
  wire::Reader r(b, tmp, compressed);
  auto u = r.vec(wire::K_G1, 2);
  auto v = r.vec(wire::K_G2, 2);
  auto g1 = r.elem(wire::K_G1);
  auto g2 = r.elem(wire::K_G2);
  auto gt = r.elem(wire::K_GT);
  r.finish();
  if (u.size() != 4 || v.size() != 4) throw SerializationError("InvalidData: the SXDH CRS has two keys per group");
  return CRS(r.coms<Com1>(u), r.coms<Com2>(v), G1Affine{r.get(g1)}, G2Affine{r.get(g2)}, GT{r.get(gt)}, curve, device);
}

// ---- several GPUs of one node (gs_ctx_create_multi): a batch of N same-shaped equations, proved / verified in
// contiguous blocks, one per device.  The reference has no batch type (Statement = Vec<dyn Equ> is unused there), so
// the batch is given as vectors of equations and per-equation witnesses; everything else follows Equation<>.
struct MultiCtx {
  gs_multi* h = nullptr;
  size_t sz[6] = {0, 0, 0, 0, 0, 0};
  MultiCtx(int curve, const std::vector<int>& devices) {
    if (gs_sizes(curve, sz) != GS_OK) throw std::runtime_error("bad curve id");
    if (gs_ctx_create_multi(curve, devices.data(), (int)devices.size(), &h) != GS_OK)
      throw std::runtime_error("gs_ctx_create_multi failed (no usable GPU; there is no CPU fallback)");
  }
  ~MultiCtx() { gs_multi_destroy(h); }
  MultiCtx(const MultiCtx&) = delete;
  MultiCtx& operator=(const MultiCtx&) = delete;
  void chk(int rc) const {
    if (rc == GS_ERR_SHAPE) throw Panic(gs_multi_last_error(h));
    if (rc != GS_OK) throw std::runtime_error(std::string("gs_amd error: ") + gs_multi_last_error(h));
  }
  void set_crs(const CRS& crs) {
    Bytes flat = cat(crs.u), t = cat(crs.v);
    flat.insert(flat.end(), t.begin(), t.end());
    flat.insert(flat.end(), crs.g1_gen.v.begin(), crs.g1_gen.v.end());
    flat.insert(flat.end(), crs.g2_gen.v.begin(), crs.g2_gen.v.end());
    flat.insert(flat.end(), crs.gt_gen.v.begin(), crs.gt_gen.v.end());
    assert_eq(flat.size(), sz[5], "CRS size");
    chk(gs_multi_set_crs(h, flat.data()));
  }
  // verify N proofs of N equations of ONE type and shape (m x n); ok[i] as Verifiable::verify of equation i
  template <class Equ> std::vector<bool> verify_batch(const std::vector<Equ>& equs, const std::vector<CProof>& proofs) {
    assert_eq(equs.size(), proofs.size(), "equs.len() == proofs.len()");
    size_t N = equs.size();
    if (N == 0) return {};
    size_t m = proofs[0].xcoms.coms.size(), n = proofs[0].ycoms.coms.size();
    if (m == 0 || n == 0) throw Panic("index out of bounds: empty commitment list");
    Bytes A, B, G, tg, xc, yc, pi, th;
    auto app = [](Bytes& d, const Bytes& s) { d.insert(d.end(), s.begin(), s.end()); };
    const EquType ty = equs[0].get_type();
    const bool xs = ty == EquType::MultiScalarG2 || ty == EquType::Quadratic, ys = ty == EquType::MultiScalarG1 || ty == EquType::Quadratic;
    const size_t sx = xs ? sz[1] : sz[2], sy = ys ? sz[1] : sz[3];
    const size_t tsz = ty == EquType::PairingProduct ? sz[4] : ty == EquType::MultiScalarG1 ? sz[2]
                       : ty == EquType::MultiScalarG2 ? sz[3] : sz[1];
    for (size_t i = 0; i < N; i++) {
      const Equ& e = equs[i];
      const CProof& p = proofs[i];
      assert_eq(p.equ_proofs.size(), 1, "com_proof.equ_proofs.len() == 1");
      assert_eq(p.xcoms.coms.size(), m, "same m for the whole batch");
      assert_eq(p.ycoms.coms.size(), n, "same n for the whole batch");
      e.check_statement_shape(m, n);
      assert_eq(p.equ_proofs[0].pi.size(), Equ::KX, "pi.len()");
      assert_eq(p.equ_proofs[0].theta.size(), Equ::KY, "theta.len()");
      // per equation, as Equation::verify does: ONE short target or constant would shift every later equation's
      // operands and let the C ABI read past the end of the buffers
      assert_eq(e.target.v.size(), tsz, "target size");
      app(A, cat(e.a_consts));
      app(B, cat(e.b_consts));
      app(G, cat(e.gamma));
      app(tg, e.target.v);
      app(xc, cat(p.xcoms.coms));
      app(yc, cat(p.ycoms.coms));
      app(pi, cat(p.equ_proofs[0].pi));
      app(th, cat(p.equ_proofs[0].theta));
    }
    assert_eq(G.size(), N * m * n * sz[1], "gamma bytes");
    assert_eq(xc.size(), N * m * 2 * sz[2], "xcoms bytes");
    assert_eq(yc.size(), N * n * 2 * sz[3], "ycoms bytes");
    assert_eq(pi.size(), N * Equ::KX * 2 * sz[3], "pi bytes");
    assert_eq(th.size(), N * Equ::KY * 2 * sz[2], "theta bytes");
    assert_eq(A.size(), N * n * sx, "a_consts bytes");
    assert_eq(B.size(), N * m * sy, "b_consts bytes");
    assert_eq(tg.size(), N * tsz, "target bytes");
    std::vector<uint8_t> ok(N, 0);
    chk(gs_multi_verify_batch(h, (int)equs[0].get_type(), N, (int)m, (int)n, A.data(), B.data(), G.data(), tg.data(),
                              xc.data(), yc.data(), pi.data(), th.data(), ok.data()));
    return std::vector<bool>(ok.begin(), ok.end());
  }
};

}  // namespace gs_amd
