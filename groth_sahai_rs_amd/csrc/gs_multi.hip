// Several GPUs of one node behind the C ABI (include/gs_amd.h, "multi-GPU" section): SURVEY.md 8b's
// gs_ctx_create(curve, devices[], ndev) contract and 8e's partitioning.
//
// Equations are independent given the shared CRS, so a batch is cut into contiguous blocks of equation indices, one per
// shard; every shard owns a full gs_ctx (stream, scratch; the CRS tables are shared by the shards of one device) and
// runs the ordinary single-device entry points on its block from its own PERSISTENT host thread (one worker per shard,
// bound to its device once; a call posts one job per worker).  Prove and exact verify need NO collective: inputs and
// outputs of a block never leave its device.  The batched (RLC) verifier produces one 1152-byte accumulator pair per
// shard, written straight into the shard's exchange buffer on its device; the cross-device reduction is a PRODUCT in
// Fp12, which is not an RCCL reduction operator, so the pairs are all-gathered (RCCL over xGMI between distinct GPUs;
// < 10 KB, latency-bound) and multiplied in shard order on device 0, followed by ONE final exponentiation.  RCCL is
// bound at run time (dlopen; the copy already in the process -- e.g. PyTorch's -- is preferred so that one HIP
// runtime serves both) and only used when the shards sit on distinct devices; one shard, shards that share a device
// (GS_MULTI_SHARED_DEVICES: how the split is tested on a one-GPU box) or a missing librccl exchange the pairs with
// device-to-device / peer copies instead -- the same bytes in the same order.
//
// Two families of entry points over the same split: host pointers (whole batch in, whole batch out; every shard stages
// its block through its context's pinned pipeline) and `_dev` (per-shard DEVICE pointer arrays: shard i's block already
// sits on devices[i], nothing crosses PCIe; asynchronous, gs_multi_sync joins).
#include "../../include/gs_amd.h"

#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>

#include <condition_variable>
#include <functional>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

namespace {

// ---- the RCCL entry points used, bound with dlsym -------------------------------------------------------
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

// one persistent host thread per shard: bound to the shard's device once, then runs the jobs posted to it
struct Worker {
  std::thread th;
  std::mutex mu;
  std::condition_variable cv;
  std::function<int()> job;
  bool has = false, done = false, stop = false;
  int rc = GS_OK;
  void start(int device) {
    th = std::thread([this, device] {
      hipSetDevice(device);
      for (;;) {
        std::function<int()> j;
        {
          std::unique_lock<std::mutex> lk(mu);
          cv.wait(lk, [this] { return has || stop; });
          if (stop) return;
          j = job;
          has = false;
        }
        int r = j();
        {
          std::lock_guard<std::mutex> lk(mu);
          rc = r;
          done = true;
        }
        cv.notify_all();
      }
    });
  }
  void post(std::function<int()> j) {
    {
      std::lock_guard<std::mutex> lk(mu);
      job = std::move(j);
      has = true;
      done = false;
    }
    cv.notify_all();
  }
  int wait() {
    std::unique_lock<std::mutex> lk(mu);
    cv.wait(lk, [this] { return done; });
    return rc;
  }
  void shutdown() {
    {
      std::lock_guard<std::mutex> lk(mu);
      stop = true;
    }
    cv.notify_all();
    if (th.joinable()) th.join();
  }
};

}  // namespace

struct gs_multi {
  int curve = 0;
  std::vector<int> devices;
  std::vector<gs_ctx*> ctx;
  std::vector<Worker*> workers;  // empty for one shard (runs inline)
  std::string err;
  bool distinct = true;  // every shard on its own device
  // accumulator exchange (batched verifier): per shard one pair out, all pairs in; RCCL where it applies
  Rccl rccl;
  std::vector<ncclComm_t> comms;
  std::vector<hipStream_t> cstream;
  std::vector<void*> sendb, recvb;
  bool bufs_ready = false, rccl_ready = false, rccl_failed = false;
  std::string rccl_why;
  // The pair exchange has three LEGS, tried in this order; a leg that fails at run time is marked failed (with its
  // reason) and the SAME call goes on with the next one -- it does not return an error while a slower way exists:
  //   0  RCCL all-gather            (shards on distinct devices, librccl loaded, communicators up)
  //   1  device / peer copies       (hipMemcpyAsync on one device, hipMemcpyPeerAsync between devices)
  //   2  host-staged copies         (every pair down to the host, the gathered block up to every shard)
  bool leg_failed[3] = {false, false, false};
  std::string leg_why[3];
  int leg_used = -1;       // what the last exchange ran on
  int fail_mask = 0;       // test hook ("exchange_fail", GS_MULTI_EXCHANGE_FAIL): bit k makes leg k fail when it runs
  int first_leg = 0;       // "exchange_leg": start the chain at this leg
  bool peer_probed = false, peer_ok = true;  // hipDeviceCanAccessPeer / hipDeviceEnablePeerAccess over all shard pairs
  std::string note;
  size_t gt = 0;
  std::vector<uint8_t> one;  // the GT identity in boundary form (accumulator pair of an empty block)
};

static int mfail(gs_multi* m, int code, const std::string& what) {
  if (m) m->err = what;
  return code;
}
// contiguous block [lo, hi) of equation indices of shard i (sizes differ by at most one)
static void block(size_t N, int ndev, int i, size_t* lo, size_t* hi) {
  size_t base = N / ndev, rem = N % ndev;
  *lo = i * base + ((size_t)i < rem ? (size_t)i : rem);
  *hi = *lo + base + ((size_t)i < rem ? 1 : 0);
}
static inline const uint8_t* off(const void* p, size_t bytes) { return p ? (const uint8_t*)p + bytes : nullptr; }
static inline uint8_t* offw(void* p, size_t bytes) { return p ? (uint8_t*)p + bytes : nullptr; }

// run fn(i) for every shard on its worker; first non-zero status wins
template <class F> static int on_all(gs_multi* m, F fn) {
  int nd = (int)m->ctx.size();
  std::vector<int> rc(nd, GS_OK);
  if (m->workers.empty()) {
    for (int i = 0; i < nd; i++) rc[i] = fn(i);
  } else {
    for (int i = 0; i < nd; i++) m->workers[i]->post([&fn, i] { return fn(i); });
    for (int i = 0; i < nd; i++) rc[i] = m->workers[i]->wait();
  }
  for (int i = 0; i < nd; i++)
    if (rc[i] != GS_OK) {
      char b[64];
      snprintf(b, sizeof b, "shard %d (device %d): ", i, m->devices[i]);
      m->err = std::string(b) + gs_last_error(m->ctx[i]);
      return rc[i];
    }
  return GS_OK;
}

static void probe_peers(gs_multi* m);
// exchange buffers (always) and, for distinct devices, the RCCL communicators.  A failure to bring RCCL up is
// remembered with its reason and the exchange falls back to peer copies; a failed buffer allocation is retried by the
// next call (nothing sticky but what succeeded).
static int exchange_setup(gs_multi* m) {
  int nd = (int)m->ctx.size();
  if (!m->bufs_ready) {
    m->cstream.resize(nd, nullptr);
    m->sendb.resize(nd, nullptr);
    m->recvb.resize(nd, nullptr);
    for (int i = 0; i < nd; i++) {
      if (hipSetDevice(m->devices[i]) != hipSuccess) return mfail(m, GS_ERR_DEVICE, "hipSetDevice");
      if (!m->cstream[i] && hipStreamCreateWithFlags(&m->cstream[i], hipStreamNonBlocking) != hipSuccess)
        return mfail(m, GS_ERR_DEVICE, "exchange stream");
      if (!m->sendb[i] && hipMalloc(&m->sendb[i], 2 * m->gt) != hipSuccess) return mfail(m, GS_ERR_ALLOC, "exchange buffers");
      if (!m->recvb[i] && hipMalloc(&m->recvb[i], (size_t)nd * 2 * m->gt) != hipSuccess)
        return mfail(m, GS_ERR_ALLOC, "exchange buffers");
    }
    m->bufs_ready = true;
  }
  probe_peers(m);
  if (nd > 1 && m->distinct && !m->rccl_ready && !m->rccl_failed) {
    m->rccl = load_rccl();
    if (!m->rccl.ok()) {
      m->rccl_failed = true;
      m->rccl_why = "librccl.so could not be loaded";
    } else {
      m->comms.assign(nd, nullptr);
      ncclResult_t r = m->rccl.CommInitAll(m->comms.data(), nd, m->devices.data());
      if (r != 0) {
        m->rccl_failed = true;
        m->rccl_why = std::string("ncclCommInitAll: ") + (m->rccl.GetErrorString ? m->rccl.GetErrorString(r) : "?");
        m->comms.clear();
      } else {
        m->rccl_ready = true;
      }
    }
  }
  return GS_OK;
}

// peer access between the shards' devices, probed once (gs_ctx_create_multi): hipMemcpyPeerAsync works without it (the
// runtime stages through the host) but with it the copy is one xGMI transfer; what was found goes into the note
static void probe_peers(gs_multi* m) {
  if (m->peer_probed) return;
  m->peer_probed = true;
  int nd = (int)m->ctx.size();
  for (int i = 0; i < nd; i++)
    for (int j = 0; j < nd; j++) {
      if (m->devices[i] == m->devices[j]) continue;
      int can = 0;
      if (hipDeviceCanAccessPeer(&can, m->devices[i], m->devices[j]) != hipSuccess || !can) {
        (void)hipGetLastError();
        m->peer_ok = false;
        continue;
      }
      if (hipSetDevice(m->devices[i]) == hipSuccess) {
        hipError_t e = hipDeviceEnablePeerAccess(m->devices[j], 0);
        if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) m->peer_ok = false;
        (void)hipGetLastError();
      }
    }
}

static bool leg_applies(const gs_multi* m, int leg) {
  if (m->leg_failed[leg]) return false;
  if (leg == 0) return m->rccl_ready || (m->fail_mask & 1);  // (the hook lets a one-GPU box walk the chain from the top)
  return true;
}
static int run_leg(gs_multi* m, int leg) {
  int nd = (int)m->ctx.size();
  const size_t pair = 2 * m->gt;
  if (m->fail_mask & (1 << leg)) return mfail(m, GS_ERR_DEVICE, "injected failure (exchange_fail)");
  if (leg == 0) {
    m->rccl.GroupStart();
    for (int i = 0; i < nd; i++) {
      ncclResult_t r = m->rccl.AllGather(m->sendb[i], m->recvb[i], pair, kNcclUint8, m->comms[i], m->cstream[i]);
      if (r != 0) {
        m->rccl.GroupEnd();
        return mfail(m, GS_ERR_DEVICE, std::string("ncclAllGather: ") + (m->rccl.GetErrorString ? m->rccl.GetErrorString(r) : "?"));
      }
    }
    if (m->rccl.GroupEnd() != 0) return mfail(m, GS_ERR_DEVICE, "ncclGroupEnd");
  } else if (leg == 1) {
    for (int i = 0; i < nd; i++) {
      if (hipSetDevice(m->devices[i]) != hipSuccess) return mfail(m, GS_ERR_DEVICE, "hipSetDevice");
      for (int j = 0; j < nd; j++) {
        uint8_t* dst = (uint8_t*)m->recvb[i] + (size_t)j * pair;
        hipError_t e = m->devices[i] == m->devices[j]
                           ? hipMemcpyAsync(dst, m->sendb[j], pair, hipMemcpyDeviceToDevice, m->cstream[i])
                           : hipMemcpyPeerAsync(dst, m->devices[i], m->sendb[j], m->devices[j], pair, m->cstream[i]);
        if (e != hipSuccess) {
          (void)hipGetLastError();
          return mfail(m, GS_ERR_DEVICE, std::string("pair exchange (device / peer copy): ") + hipGetErrorString(e));
        }
      }
    }
  } else {
    // host-staged: nd small downloads, nd uploads of the gathered block (nd x 1152 bytes); synchronous copies
    std::vector<uint8_t> all((size_t)nd * pair);
    for (int j = 0; j < nd; j++)
      if (hipSetDevice(m->devices[j]) != hipSuccess ||
          hipMemcpy(all.data() + (size_t)j * pair, m->sendb[j], pair, hipMemcpyDeviceToHost) != hipSuccess) {
        (void)hipGetLastError();
        return mfail(m, GS_ERR_DEVICE, "pair exchange (host-staged download)");
      }
    for (int i = 0; i < nd; i++)
      if (hipSetDevice(m->devices[i]) != hipSuccess ||
          hipMemcpy(m->recvb[i], all.data(), all.size(), hipMemcpyHostToDevice) != hipSuccess) {
        (void)hipGetLastError();
        return mfail(m, GS_ERR_DEVICE, "pair exchange (host-staged upload)");
      }
    return GS_OK;
  }
  for (int i = 0; i < nd; i++)
    if (hipSetDevice(m->devices[i]) != hipSuccess || hipStreamSynchronize(m->cstream[i]) != hipSuccess) {
      (void)hipGetLastError();
      return mfail(m, GS_ERR_DEVICE, leg == 0 ? "rccl all-gather (sync)" : "pair exchange (sync)");
    }
  return GS_OK;
}

// every shard's pair sits in sendb[i] (its context's stream has been drained): all nd pairs, in shard order, into
// every recvb[i].  Walks the legs (see gs_multi): a failed leg is remembered and the call retries on the next one.
static int exchange_pairs(gs_multi* m) {
  static const char* names[3] = {"rccl all-gather", "device / peer copies", "host-staged copies"};
  int nd = (int)m->ctx.size();
  std::string chain;
  for (int leg = m->first_leg; leg < 3; leg++) {
    if (!leg_applies(m, leg)) continue;
    int rc = run_leg(m, leg);
    if (rc == GS_OK) {
      m->leg_used = leg;
      return GS_OK;
    }
    m->leg_failed[leg] = true;
    m->leg_why[leg] = m->err;
    chain += std::string(names[leg]) + " failed (" + m->err + "); ";
    // whatever the failed leg left queued must not land in the buffers while the next leg fills them
    for (int i = 0; i < nd; i++)
      if (hipSetDevice(m->devices[i]) == hipSuccess && m->cstream[i]) (void)hipStreamSynchronize(m->cstream[i]);
    (void)hipGetLastError();
  }
  return mfail(m, GS_ERR_DEVICE, "pair exchange: every leg failed: " + chain);
}

// after the shards wrote their pairs: gather, cross-check, one final exponentiation on device 0
static int finish_rlc(gs_multi* m, void* acc_pairs, uint8_t* ok_all) {
  int nd = (int)m->ctx.size();
  const size_t pair = 2 * m->gt;
  int rc = exchange_pairs(m);
  if (rc != GS_OK) return rc;
  // every shard must have received the same bytes (a wrong rank order or a torn transfer would change the verdict)
  std::vector<uint8_t> gathered((size_t)nd * pair), check((size_t)nd * pair);
  for (int i = 0; i < nd; i++) {
    if (hipSetDevice(m->devices[i]) != hipSuccess ||
        hipMemcpy(i == 0 ? gathered.data() : check.data(), m->recvb[i], (size_t)nd * pair, hipMemcpyDeviceToHost) != hipSuccess)
      return mfail(m, GS_ERR_DEVICE, "accumulator download");
    if (i > 0 && memcmp(check.data(), gathered.data(), gathered.size()) != 0)
      return mfail(m, GS_ERR_DEVICE, "gathered accumulator pairs differ between shards");
  }
  if (acc_pairs) memcpy(acc_pairs, gathered.data(), gathered.size());
  // product in shard order + ONE final exponentiation, on device 0, from its own receive buffer
  rc = gs_gt_finalize_dev(m->ctx[0], (size_t)nd, m->recvb[0], ok_all);
  if (rc != GS_OK) return mfail(m, rc, gs_last_error(m->ctx[0]));
  return GS_OK;
}

extern "C" {

int gs_ctx_create_multi_ex(int curve, const int* devices, int ndev, int flags, gs_multi** out) {
  if (!out || !devices || ndev < 1 || ndev > 64 || (curve != 0 && curve != 1)) return GS_ERR_ARG;
  *out = nullptr;
  bool distinct = true;
  for (int i = 0; i < ndev; i++)
    for (int k = 0; k < i; k++)
      if (devices[i] == devices[k]) distinct = false;
  if (!distinct && !(flags & GS_MULTI_SHARED_DEVICES)) return GS_ERR_ARG;  // one context per device unless asked for
  gs_multi* m = new gs_multi();
  m->curve = curve;
  m->distinct = distinct;
  size_t sz[6];
  gs_sizes(curve, sz);
  m->gt = sz[4];
  for (int i = 0; i < ndev; i++) {
    gs_ctx* c = nullptr;
    int rc = gs_ctx_create(curve, devices[i], &c);
    if (rc != GS_OK) {
      gs_multi_destroy(m);
      return rc;
    }
    m->devices.push_back(devices[i]);
    m->ctx.push_back(c);
  }
  if (ndev > 1)
    for (int i = 0; i < ndev; i++) {
      m->workers.push_back(new Worker());
      m->workers.back()->start(devices[i]);
    }
  if (const char* e = getenv("GS_MULTI_EXCHANGE_FAIL")) m->fail_mask = atoi(e) & 7;
  if (const char* e = getenv("GS_MULTI_EXCHANGE_LEG")) m->first_leg = std::min(2, std::max(0, atoi(e)));
  probe_peers(m);
  hipSetDevice(devices[0]);
  *out = m;
  return GS_OK;
}
int gs_ctx_create_multi(int curve, const int* devices, int ndev, gs_multi** out) {
  return gs_ctx_create_multi_ex(curve, devices, ndev, 0, out);
}

void gs_multi_destroy(gs_multi* m) {
  if (!m) return;
  for (Worker* w : m->workers) {
    w->shutdown();
    delete w;
  }
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
int gs_multi_uses_rccl(gs_multi* m) { return (m && m->rccl_ready && !m->leg_failed[0]) ? 1 : 0; }
const char* gs_multi_exchange_note(gs_multi* m) {
  if (!m) return "null context";
  static const char* names[3] = {"rccl all-gather", "device / peer copies", "host-staged copies"};
  std::string n;
  if (m->leg_used >= 0) {
    n = std::string("last exchange: ") + names[m->leg_used];
    if (m->leg_used == 1 && m->ctx.size() == 1) n += " (one shard: device-to-device copy)";
    if (m->leg_used == 1 && m->ctx.size() > 1 && !m->distinct) n += " (shards share a device)";
  }
  else if (m->rccl_ready && !m->leg_failed[0])
    n = "rccl all-gather";
  else if (m->ctx.size() == 1)
    n = "one shard: device-to-device copy";
  else if (!m->distinct)
    n = "shards share a device: device-to-device copies";
  else if (m->rccl_failed)
    n = "device / peer copies (" + m->rccl_why + ")";
  else
    n = "not set up yet";
  for (int k = 0; k < 3; k++)
    if (m->leg_failed[k]) n += std::string("; ") + names[k] + " failed earlier: " + m->leg_why[k];
  if (m->peer_probed && m->distinct && m->ctx.size() > 1) n += m->peer_ok ? "; peer access enabled" : "; no peer access between some shards";
  m->note = n;
  return m->note.c_str();
}
// "exchange_leg" 0..2: the leg the pair exchange starts at (0 = RCCL where it applies); "exchange_fail": bit k makes
// leg k fail when it runs (test hook for the fallback chain); "exchange_reset" (any value): forget earlier failures
int gs_multi_set_option(gs_multi* m, const char* key, int value) {
  if (!m || !key) return GS_ERR_ARG;
  std::string k(key);
  if (k == "exchange_leg") {
    if (value < 0 || value > 2) return mfail(m, GS_ERR_ARG, "exchange_leg: 0 rccl, 1 device / peer copies, 2 host-staged");
    m->first_leg = value;
  } else if (k == "exchange_fail") {
    if (value < 0 || value > 7) return mfail(m, GS_ERR_ARG, "exchange_fail: bit mask of legs 0..2");
    m->fail_mask = value;
  } else if (k == "exchange_reset") {
    for (int i = 0; i < 3; i++) m->leg_failed[i] = false, m->leg_why[i].clear();
    m->leg_used = -1;
  } else {
    return mfail(m, GS_ERR_ARG, "unknown option");
  }
  return GS_OK;
}

int gs_multi_shard(gs_multi* m, size_t N, int i, size_t* lo, size_t* hi) {
  if (!m || !lo || !hi || i < 0 || (size_t)i >= m->ctx.size()) return GS_ERR_ARG;
  block(N, (int)m->ctx.size(), i, lo, hi);
  return GS_OK;
}

int gs_multi_set_crs(gs_multi* m, const void* crs) {
  if (!m || !crs) return GS_ERR_ARG;
  // (shards of one device: the first builds the tables, the others find and share them)
  if (!m->distinct) {
    for (size_t i = 0; i < m->ctx.size(); i++) {
      int rc = gs_set_crs(m->ctx[i], crs);
      if (rc != GS_OK) return mfail(m, rc, gs_last_error(m->ctx[i]));
    }
    return GS_OK;
  }
  return on_all(m, [&](int i) { return gs_set_crs(m->ctx[i], crs); });
}

int gs_multi_sync(gs_multi* m) {
  if (!m) return GS_ERR_ARG;
  return on_all(m, [&](int i) { return gs_sync(m->ctx[i]); });
}

// ---- host pointers: whole batch in, whole batch out ------------------------------------------------------------
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

static int rlc_prologue(gs_multi* m, int ty, size_t N, int mm, int n, Shape* s) {
  if (!shape_of(m->curve, ty, s)) return mfail(m, GS_ERR_ARG, "bad equation type");
  if (mm < 1 || n < 1) return mfail(m, GS_ERR_SHAPE, "m and n must be >= 1");
  if (N == 0) return mfail(m, GS_ERR_ARG, "empty batch");
  int rc = exchange_setup(m);
  if (rc != GS_OK) return rc;
  if (m->one.empty()) {  // the accumulator pair of an empty block is (1, 1): E::multi_pairing of no pairs
    m->one.resize(m->gt);
    rc = gs_multi_pairing_batch(m->ctx[0], 1, 0, m->one.data(), m->one.data(), m->one.data());
    if (rc != GS_OK) {
      m->one.clear();
      return mfail(m, rc, gs_last_error(m->ctx[0]));
    }
  }
  return GS_OK;
}
static int empty_pair(gs_multi* m, int i) {
  if (hipMemcpy(m->sendb[i], m->one.data(), m->gt, hipMemcpyHostToDevice) != hipSuccess ||
      hipMemcpy((uint8_t*)m->sendb[i] + m->gt, m->one.data(), m->gt, hipMemcpyHostToDevice) != hipSuccess)
    return GS_ERR_DEVICE;
  return GS_OK;
}

int gs_multi_verify_batch_rlc(gs_multi* m, int ty, size_t N, int mm, int n, const void* A, const void* B, const void* G,
                              const void* target, const void* xcoms, const void* ycoms, const void* pi,
                              const void* theta, const uint64_t* rho, void* acc_pairs, uint8_t* ok_all) {
  if (!m || !ok_all) return GS_ERR_ARG;
  if (!rho) return mfail(m, GS_ERR_ARG, "rho is NULL (see the rho contract in gs_amd.h)");
  Shape s;
  int rc = rlc_prologue(m, ty, N, mm, n, &s);
  if (rc != GS_OK) return rc;
  int nd = (int)m->ctx.size();
  const size_t pair = 2 * m->gt;
  rc = on_all(m, [&](int i) {
    size_t lo, hi;
    block(N, nd, i, &lo, &hi);
    if (hi == lo) return empty_pair(m, i);
    size_t um = (size_t)mm, un = (size_t)n;
    // (host inputs: the shard's pair comes back with them and goes up into its exchange buffer, 1152 bytes)
    std::vector<uint8_t> acc(pair);
    int r = gs_verify_batch_rlc(m->ctx[i], ty, hi - lo, mm, n, off(A, lo * un * s.sx), off(B, lo * um * s.sy),
                                off(G, lo * um * un * s.fr), off(target, lo * s.st), off(xcoms, lo * um * 4 * s.fq),
                                off(ycoms, lo * un * 8 * s.fq), off(pi, lo * s.kx * 8 * s.fq),
                                off(theta, lo * s.ky * 4 * s.fq), rho + 4 * lo, acc.data(), nullptr);
    if (r != GS_OK) return r;
    return hipMemcpy(m->sendb[i], acc.data(), pair, hipMemcpyHostToDevice) == hipSuccess ? (int)GS_OK : (int)GS_ERR_DEVICE;
  });
  if (rc != GS_OK) return rc;
  return finish_rlc(m, acc_pairs, ok_all);
}

// ---- device-resident shards: per-shard pointer arrays, nothing crosses PCIe --------------------------------------
#define PP(a) ((a) ? (a)[i] : nullptr)
int gs_multi_prove_batch_dev(gs_multi* m, int ty, size_t N, int mm, int n, const void* const* X, const void* const* Y,
                             const void* const* A, const void* const* B, const void* const* G, const void* const* R,
                             const void* const* S, const void* const* T, void* const* xcoms, void* const* ycoms,
                             void* const* pi, void* const* theta) {
  if (!m) return GS_ERR_ARG;
  if (!X || !Y || !A || !B || !G || !R || !S || !T || !pi || !theta) return mfail(m, GS_ERR_ARG, "null pointer array");
  int nd = (int)m->ctx.size();
  return on_all(m, [&](int i) {
    size_t lo, hi;
    block(N, nd, i, &lo, &hi);
    if (hi == lo) return (int)GS_OK;
    return gs_prove_batch_dev(m->ctx[i], ty, hi - lo, mm, n, X[i], Y[i], A[i], B[i], G[i], R[i], S[i], T[i], PP(xcoms),
                              PP(ycoms), pi[i], theta[i]);
  });
}
int gs_multi_verify_batch_dev(gs_multi* m, int ty, size_t N, int mm, int n, const void* const* A, const void* const* B,
                              const void* const* G, const void* const* target, const void* const* xcoms,
                              const void* const* ycoms, const void* const* pi, const void* const* theta,
                              uint8_t* const* ok) {
  if (!m) return GS_ERR_ARG;
  if (!A || !B || !G || !target || !xcoms || !ycoms || !pi || !theta || !ok) return mfail(m, GS_ERR_ARG, "null pointer array");
  int nd = (int)m->ctx.size();
  return on_all(m, [&](int i) {
    size_t lo, hi;
    block(N, nd, i, &lo, &hi);
    if (hi == lo) return (int)GS_OK;
    return gs_verify_batch_dev(m->ctx[i], ty, hi - lo, mm, n, A[i], B[i], G[i], target[i], xcoms[i], ycoms[i], pi[i],
                               theta[i], ok[i]);
  });
}
// rho[i]: shard i's exponents on ITS device, 4 per equation of its block (the caller cuts the global rho array with
// gs_multi_shard).  The shard's pair is written by the engine straight into its exchange buffer.
int gs_multi_verify_batch_rlc_dev(gs_multi* m, int ty, size_t N, int mm, int n, const void* const* A,
                                  const void* const* B, const void* const* G, const void* const* target,
                                  const void* const* xcoms, const void* const* ycoms, const void* const* pi,
                                  const void* const* theta, const uint64_t* const* rho, void* acc_pairs,
                                  uint8_t* ok_all) {
  if (!m || !ok_all) return GS_ERR_ARG;
  if (!A || !B || !G || !target || !xcoms || !ycoms || !pi || !theta || !rho) return mfail(m, GS_ERR_ARG, "null pointer array");
  Shape s;
  int rc = rlc_prologue(m, ty, N, mm, n, &s);
  if (rc != GS_OK) return rc;
  int nd = (int)m->ctx.size();
  rc = on_all(m, [&](int i) {
    size_t lo, hi;
    block(N, nd, i, &lo, &hi);
    if (hi == lo) return empty_pair(m, i);
    int r = gs_verify_batch_rlc_dev(m->ctx[i], ty, hi - lo, mm, n, A[i], B[i], G[i], target[i], xcoms[i], ycoms[i], pi[i],
                                    theta[i], rho[i], m->sendb[i]);
    return r != GS_OK ? r : gs_sync(m->ctx[i]);
  });
  if (rc != GS_OK) return rc;
  return finish_rlc(m, acc_pairs, ok_all);
}
#undef PP

}  // extern "C"
