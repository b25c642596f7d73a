// Several GPUs of one node behind the C ABI (include/gs_amd.h, "multi-GPU" section): SURVEY.md 8b's
// gs_ctx_create(curve, devices[], ndev) contract and 8e's partitioning.
//
// Equations are independent given the shared CRS, so a batch is cut into contiguous blocks of equation indices, one per
// device; every device owns a full gs_ctx (stream, scratch, its own copy of the CRS tables) and runs the ordinary
// single-device entry points on its block from its own host thread.  Prove and exact verify need NO collective: inputs
// and outputs of a block never leave its device.  The batched (RLC) verifier produces one 1152-byte accumulator pair
// per device; the cross-device reduction is a PRODUCT in Fp12, which is not an RCCL reduction operator, so the pairs
// are all-gathered over RCCL (xGMI between the GPUs of a node; < 10 KB, latency-bound) and multiplied in device order,
// followed by ONE final exponentiation.  RCCL is bound at run time (dlopen; the copy already in the process -- e.g.
// PyTorch's -- is preferred so that one HIP runtime serves both) and only needed for that exchange.
#include "../../include/gs_amd.h"

#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>

#include <string>
#include <thread>
#include <vector>

namespace {

// ---- the five RCCL entry points used, bound with dlsym -------------------------------------------------
typedef struct ncclComm* ncclComm_t;
typedef int ncclResult_t;  // ncclSuccess == 0
enum { kNcclUint8 = 1 };   // ncclDataType_t: ncclInt8 = 0, ncclUint8 = 1 (rccl.h)
struct Rccl {
  void* h = nullptr;
  ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  ncclResult_t (*AllGather)(const void*, void*, size_t, int, ncclComm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  bool ok() const { return h && CommInitAll && CommDestroy && GroupStart && GroupEnd && AllGather; }
};
static Rccl load_rccl() {
  Rccl r;
  const char* names[] = {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so.1"};
  for (const char* n : names)  // a copy that is already mapped (PyTorch's) first: it matches the HIP runtime in use
    if ((r.h = dlopen(n, RTLD_NOW | RTLD_NOLOAD | RTLD_LOCAL))) break;
  for (const char* n : names) {
    if (r.h) break;
    r.h = dlopen(n, RTLD_NOW | RTLD_LOCAL);
  }
  if (!r.h) return r;
  r.CommInitAll = (decltype(r.CommInitAll))dlsym(r.h, "ncclCommInitAll");
  r.CommDestroy = (decltype(r.CommDestroy))dlsym(r.h, "ncclCommDestroy");
  r.GroupStart = (decltype(r.GroupStart))dlsym(r.h, "ncclGroupStart");
  r.GroupEnd = (decltype(r.GroupEnd))dlsym(r.h, "ncclGroupEnd");
  r.AllGather = (decltype(r.AllGather))dlsym(r.h, "ncclAllGather");
  r.GetErrorString = (decltype(r.GetErrorString))dlsym(r.h, "ncclGetErrorString");
  return r;
}

struct Shape {
  size_t fq, fr, sx, sy, st;
  int kx, ky;
};
static bool shape_of(int curve, int ty, Shape* s) {
  size_t sz[6];
  if (ty < 0 || ty > 3 || gs_sizes(curve, sz) != GS_OK) return false;
  bool xg = ty == GS_PPE || ty == GS_MSMEG1, yg = ty == GS_PPE || ty == GS_MSMEG2;
  s->fq = sz[0];
  s->fr = sz[1];
  s->kx = xg ? 2 : 1;
  s->ky = yg ? 2 : 1;
  s->sx = xg ? sz[2] : sz[1];
  s->sy = yg ? sz[3] : sz[1];
  s->st = ty == GS_PPE ? sz[4] : ty == GS_MSMEG1 ? sz[2] : ty == GS_MSMEG2 ? sz[3] : sz[1];
  return true;
}

}  // namespace

struct gs_multi {
  int curve = 0;
  std::vector<int> devices;
  std::vector<gs_ctx*> ctx;
  std::string err;
  // RCCL (bound lazily; only the batched verifier's accumulator exchange needs it)
  Rccl rccl;
  std::vector<ncclComm_t> comms;
  std::vector<hipStream_t> cstream;
  std::vector<void*> sendb, recvb;  // per device: one pair in, ndev pairs out
  bool rccl_tried = false, rccl_ready = false;
  size_t gt = 0;
};

static int mfail(gs_multi* m, int code, const std::string& what) {
  if (m) m->err = what;
  return code;
}
// contiguous block [lo, hi) of equation indices of device i (sizes differ by at most one)
static void block(size_t N, int ndev, int i, size_t* lo, size_t* hi) {
  size_t base = N / ndev, rem = N % ndev;
  *lo = i * base + ((size_t)i < rem ? (size_t)i : rem);
  *hi = *lo + base + ((size_t)i < rem ? 1 : 0);
}
static inline const uint8_t* off(const void* p, size_t bytes) { return p ? (const uint8_t*)p + bytes : nullptr; }
static inline uint8_t* offw(void* p, size_t bytes) { return p ? (uint8_t*)p + bytes : nullptr; }

// run fn(i) for every device on its own host thread; first non-zero status wins
template <class F> static int on_all(gs_multi* m, F fn) {
  int nd = (int)m->ctx.size();
  std::vector<int> rc(nd, GS_OK);
  if (nd == 1) {
    rc[0] = fn(0);
  } else {
    std::vector<std::thread> th;
    for (int i = 0; i < nd; i++) th.emplace_back([&, i] { rc[i] = fn(i); });
    for (auto& t : th) t.join();
  }
  for (int i = 0; i < nd; i++)
    if (rc[i] != GS_OK) {
      char b[64];
      snprintf(b, sizeof b, "device %d: ", m->devices[i]);
      m->err = std::string(b) + gs_last_error(m->ctx[i]);
      return rc[i];
    }
  return GS_OK;
}

static int rccl_setup(gs_multi* m) {
  if (m->rccl_tried) return m->rccl_ready ? GS_OK : GS_ERR_DEVICE;
  m->rccl_tried = true;
  m->rccl = load_rccl();
  if (!m->rccl.ok()) return mfail(m, GS_ERR_DEVICE, "RCCL (librccl.so) could not be loaded");
  int nd = (int)m->ctx.size();
  m->comms.assign(nd, nullptr);
  ncclResult_t r = m->rccl.CommInitAll(m->comms.data(), nd, m->devices.data());
  if (r != 0)
    return mfail(m, GS_ERR_DEVICE, std::string("ncclCommInitAll: ") + (m->rccl.GetErrorString ? m->rccl.GetErrorString(r) : "?"));
  m->cstream.assign(nd, nullptr);
  m->sendb.assign(nd, nullptr);
  m->recvb.assign(nd, nullptr);
  for (int i = 0; i < nd; i++) {
    if (hipSetDevice(m->devices[i]) != hipSuccess || hipStreamCreateWithFlags(&m->cstream[i], hipStreamNonBlocking) != hipSuccess ||
        hipMalloc(&m->sendb[i], 2 * m->gt) != hipSuccess || hipMalloc(&m->recvb[i], (size_t)nd * 2 * m->gt) != hipSuccess)
      return mfail(m, GS_ERR_ALLOC, "RCCL exchange buffers");
  }
  m->rccl_ready = true;
  return GS_OK;
}

extern "C" {

int gs_ctx_create_multi(int curve, const int* devices, int ndev, gs_multi** out) {
  if (!out || !devices || ndev < 1 || ndev > 64 || (curve != 0 && curve != 1)) return GS_ERR_ARG;
  *out = nullptr;
  for (int i = 0; i < ndev; i++)
    for (int k = 0; k < i; k++)
      if (devices[i] == devices[k]) return GS_ERR_ARG;  // one context per device
  gs_multi* m = new gs_multi();
  m->curve = curve;
  size_t sz[6];
  gs_sizes(curve, sz);
  m->gt = sz[4];
  for (int i = 0; i < ndev; i++) {
    gs_ctx* c = nullptr;
    int rc = gs_ctx_create(curve, devices[i], &c);
    if (rc != GS_OK) {
      for (gs_ctx* p : m->ctx) gs_ctx_destroy(p);
      delete m;
      return rc;
    }
    m->devices.push_back(devices[i]);
    m->ctx.push_back(c);
  }
  *out = m;
  return GS_OK;
}

void gs_multi_destroy(gs_multi* m) {
  if (!m) return;
  for (size_t i = 0; i < m->ctx.size(); i++) {
    hipSetDevice(m->devices[i]);
    if (i < m->cstream.size() && m->cstream[i]) {
      hipStreamSynchronize(m->cstream[i]);
      hipStreamDestroy(m->cstream[i]);
    }
    if (i < m->sendb.size() && m->sendb[i]) hipFree(m->sendb[i]);
    if (i < m->recvb.size() && m->recvb[i]) hipFree(m->recvb[i]);
    if (i < m->comms.size() && m->comms[i] && m->rccl.CommDestroy) m->rccl.CommDestroy(m->comms[i]);
  }
  for (gs_ctx* c : m->ctx) gs_ctx_destroy(c);
  delete m;
}

int gs_multi_ndev(gs_multi* m) { return m ? (int)m->ctx.size() : 0; }
gs_ctx* gs_multi_ctx(gs_multi* m, int i) { return (m && i >= 0 && (size_t)i < m->ctx.size()) ? m->ctx[i] : nullptr; }
const char* gs_multi_last_error(gs_multi* m) { return m ? m->err.c_str() : "null context"; }
int gs_multi_uses_rccl(gs_multi* m) { return (m && m->rccl_ready) ? 1 : 0; }

int gs_multi_shard(gs_multi* m, size_t N, int i, size_t* lo, size_t* hi) {
  if (!m || !lo || !hi || i < 0 || (size_t)i >= m->ctx.size()) return GS_ERR_ARG;
  block(N, (int)m->ctx.size(), i, lo, hi);
  return GS_OK;
}

int gs_multi_set_crs(gs_multi* m, const void* crs) {
  if (!m || !crs) return GS_ERR_ARG;
  return on_all(m, [&](int i) { return gs_set_crs(m->ctx[i], crs); });
}

int gs_multi_prove_batch(gs_multi* m, int ty, size_t N, int mm, int n, const void* X, const void* Y, const void* A,
                         const void* B, const void* G, const void* R, const void* S, const void* T, void* xcoms,
                         void* ycoms, void* pi, void* theta) {
  if (!m) return GS_ERR_ARG;
  Shape s;
  if (!shape_of(m->curve, ty, &s)) return mfail(m, GS_ERR_ARG, "bad equation type");
  if (mm < 1 || n < 1) return mfail(m, GS_ERR_SHAPE, "m and n must be >= 1 (reference asserts, prove.rs:106-113)");
  int nd = (int)m->ctx.size();
  return on_all(m, [&](int i) {
    size_t lo, hi;
    block(N, nd, i, &lo, &hi);
    if (hi == lo) return (int)GS_OK;
    size_t um = (size_t)mm, un = (size_t)n;
    return gs_prove_batch(m->ctx[i], ty, hi - lo, mm, n, off(X, lo * um * s.sx), off(Y, lo * un * s.sy),
                          off(A, lo * un * s.sx), off(B, lo * um * s.sy), off(G, lo * um * un * s.fr),
                          off(R, lo * um * s.kx * s.fr), off(S, lo * un * s.ky * s.fr), off(T, lo * s.ky * s.kx * s.fr),
                          offw(xcoms, lo * um * 4 * s.fq), offw(ycoms, lo * un * 8 * s.fq), offw(pi, lo * s.kx * 8 * s.fq),
                          offw(theta, lo * s.ky * 4 * s.fq));
  });
}

int gs_multi_verify_batch(gs_multi* m, int ty, size_t N, int mm, int n, const void* A, const void* B, const void* G,
                          const void* target, const void* xcoms, const void* ycoms, const void* pi, const void* theta,
                          uint8_t* ok) {
  if (!m) return GS_ERR_ARG;
  Shape s;
  if (!shape_of(m->curve, ty, &s)) return mfail(m, GS_ERR_ARG, "bad equation type");
  if (mm < 1 || n < 1) return mfail(m, GS_ERR_SHAPE, "m and n must be >= 1");
  int nd = (int)m->ctx.size();
  return on_all(m, [&](int i) {
    size_t lo, hi;
    block(N, nd, i, &lo, &hi);
    if (hi == lo) return (int)GS_OK;
    size_t um = (size_t)mm, un = (size_t)n;
    return gs_verify_batch(m->ctx[i], ty, hi - lo, mm, n, off(A, lo * un * s.sx), off(B, lo * um * s.sy),
                           off(G, lo * um * un * s.fr), off(target, lo * s.st), off(xcoms, lo * um * 4 * s.fq),
                           off(ycoms, lo * un * 8 * s.fq), off(pi, lo * s.kx * 8 * s.fq), off(theta, lo * s.ky * 4 * s.fq),
                           ok ? ok + lo : nullptr);
  });
}

int gs_multi_verify_batch_rlc(gs_multi* m, int ty, size_t N, int mm, int n, const void* A, const void* B, const void* G,
                              const void* target, const void* xcoms, const void* ycoms, const void* pi,
                              const void* theta, const uint64_t* rho, void* acc_pairs, uint8_t* ok_all) {
  if (!m || !ok_all) return GS_ERR_ARG;
  Shape s;
  if (!shape_of(m->curve, ty, &s)) return mfail(m, GS_ERR_ARG, "bad equation type");
  if (mm < 1 || n < 1) return mfail(m, GS_ERR_SHAPE, "m and n must be >= 1");
  if (N == 0) return mfail(m, GS_ERR_ARG, "empty batch");
  int nd = (int)m->ctx.size();
  const size_t pair = 2 * m->gt;
  // the accumulator pair of an empty block is (1, 1): E::multi_pairing of no pairs
  std::vector<uint8_t> one(m->gt);
  {
    int rc = gs_multi_pairing_batch(m->ctx[0], 1, 0, one.data(), one.data(), one.data());
    if (rc != GS_OK) return mfail(m, rc, gs_last_error(m->ctx[0]));
  }
  std::vector<uint8_t> local((size_t)nd * pair);
  int rc = on_all(m, [&](int i) {
    size_t lo, hi;
    block(N, nd, i, &lo, &hi);
    uint8_t* acc = &local[(size_t)i * pair];
    if (hi == lo) {
      memcpy(acc, one.data(), m->gt);
      memcpy(acc + m->gt, one.data(), m->gt);
      return (int)GS_OK;
    }
    size_t um = (size_t)mm, un = (size_t)n;
    return gs_verify_batch_rlc(m->ctx[i], ty, hi - lo, mm, n, off(A, lo * un * s.sx), off(B, lo * um * s.sy),
                               off(G, lo * um * un * s.fr), off(target, lo * s.st), off(xcoms, lo * um * 4 * s.fq),
                               off(ycoms, lo * un * 8 * s.fq), off(pi, lo * s.kx * 8 * s.fq),
                               off(theta, lo * s.ky * 4 * s.fq), rho + 4 * lo, acc, nullptr);
  });
  if (rc != GS_OK) return rc;
  // all-gather of the pairs over RCCL: every device ends with all nd pairs in device order
  std::vector<uint8_t> gathered((size_t)nd * pair);
  rc = rccl_setup(m);
  if (rc != GS_OK) return rc;
  for (int i = 0; i < nd; i++) {
    if (hipSetDevice(m->devices[i]) != hipSuccess ||
        hipMemcpyAsync(m->sendb[i], &local[(size_t)i * pair], pair, hipMemcpyHostToDevice, m->cstream[i]) != hipSuccess)
      return mfail(m, GS_ERR_DEVICE, "accumulator upload");
  }
  m->rccl.GroupStart();
  for (int i = 0; i < nd; i++) {
    ncclResult_t r = m->rccl.AllGather(m->sendb[i], m->recvb[i], pair, kNcclUint8, m->comms[i], m->cstream[i]);
    if (r != 0) {
      m->rccl.GroupEnd();
      return mfail(m, GS_ERR_DEVICE, std::string("ncclAllGather: ") + (m->rccl.GetErrorString ? m->rccl.GetErrorString(r) : "?"));
    }
  }
  if (m->rccl.GroupEnd() != 0) return mfail(m, GS_ERR_DEVICE, "ncclGroupEnd");
  std::vector<uint8_t> check((size_t)nd * pair);
  for (int i = 0; i < nd; i++) {
    if (hipSetDevice(m->devices[i]) != hipSuccess ||
        hipMemcpyAsync(i == 0 ? gathered.data() : check.data(), m->recvb[i], (size_t)nd * pair, hipMemcpyDeviceToHost,
                       m->cstream[i]) != hipSuccess ||
        hipStreamSynchronize(m->cstream[i]) != hipSuccess)
      return mfail(m, GS_ERR_DEVICE, "accumulator download");
    if (i > 0 && memcmp(check.data(), gathered.data(), gathered.size()) != 0)
      return mfail(m, GS_ERR_DEVICE, "all-gather results differ between devices");
  }
  if (memcmp(gathered.data(), local.data(), local.size()) != 0)
    return mfail(m, GS_ERR_DEVICE, "all-gather did not return the pairs in device order");
  if (acc_pairs) memcpy(acc_pairs, gathered.data(), gathered.size());
  // product in device order + ONE final exponentiation (device 0)
  rc = gs_gt_finalize(m->ctx[0], (size_t)nd, gathered.data(), ok_all);
  if (rc != GS_OK) return mfail(m, rc, gs_last_error(m->ctx[0]));
  return GS_OK;
}

}  // extern "C"
