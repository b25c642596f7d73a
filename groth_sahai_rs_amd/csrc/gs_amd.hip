// Host side of the C ABI declared in include/gs_amd.h: context, CRS tables,
// per-shape task tables ("plans"), kernel launches.  No CPU fallback: every
// compute entry point needs a working HIP device.
#include "../../include/gs_amd.h"

#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <dlfcn.h>
#include <sys/mman.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <map>
#include <memory>
#include <mutex>
#include <queue>
#include <string>
#include <thread>
#include <type_traits>
#include <unordered_map>
#include <vector>

#include "gs_params_bls12_381.h"
#include "gs_params_bn254.h"
#include "gs_kernels.cuh"

namespace gs {
GS_ZERO_ONE(Bls12_381)
GS_ZERO_ONE(Bn254)
}  // namespace gs
using namespace gs;

#define GS_VERSION "gs-amd 0.1 (gfx950)"

// ---------------------------------------------------------------------------
struct DevBuf {
  void* p = nullptr;
  size_t cap = 0;
};
struct ProfEntry {
  double ms = 0;
  uint64_t n = 0;
  uint64_t lanes = 0;  // lane-tasks launched
  uint64_t work = 0;   // kernel-specific work items (scalars, slots, pairs ...; = lanes when no hint was given)
  // clock stamps of the segmented launches (k_seg): sums over their waves of d s_memtime / d s_memrealtime
  double cyc = 0, rt = 0;
};

// Everything derived from one CRS on one device: immutable once built, shared (reference-counted) by every context of
// this process that installs the same CRS bytes on the same device -- Rayon-style callers keep one context per worker
// and would otherwise hold 1.8 GB of identical 16-bit window tables each.
struct CrsTables {
  int device = 0, curve = 0;
  std::vector<uint8_t> crs_bytes;
  DevBuf crs_g1;   // 6 G1 points: u0.0 u0.1 u1.0 u1.1 W1.0 W1.1
  DevBuf crs_g2;   // 6 G2 points: v0.0 v0.1 v1.0 v1.1 W2.0 W2.1
  DevBuf tab16_g1, tab16_g2;  // 16-bit window tables (k_build_tables16), what k_fix reads
  DevBuf line_tab;  // Miller line tables of the 6 CRS G2 points (k_line_tables)
  ~CrsTables() {
    hipSetDevice(device);
    for (DevBuf* b : {&crs_g1, &crs_g2, &tab16_g1, &tab16_g2, &line_tab})
      if (b->p) hipFree(b->p);
  }
};
static std::mutex g_tables_mu;
static std::vector<std::weak_ptr<CrsTables>> g_tables;

struct PlanEntry {  // an uploaded task table: device copy + the host bytes it was made from
  DevBuf dev;
  std::vector<uint8_t> host;
  uint64_t last_use = 0;
};

struct gs_ctx {
  int curve = 0;
  int device = 0;
  hipStream_t stream = nullptr;
  bool have_crs = false;
  std::string err;
  std::shared_ptr<CrsTables> tabs;  // CRS-derived device data (shared between contexts, see CrsTables)
  // scratch
  std::map<std::string, DevBuf> scratch;
  std::map<std::string, PlanEntry> plans;  // task tables by (name, size, content hash); content compared on a hit
  uint64_t plan_clock = 0;
  std::map<std::string, std::pair<int, double>> miller_choice;  // planner decisions by (shape, N, overrides)
  // profiling
  bool prof = false;
  std::map<std::string, ProfEntry> prof_map;
  std::vector<std::string> prof_order;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  uint64_t work_hint = 0;  // consumed by the next launch() while profiling
  // Internal side streams: the fixed-base and variable-base kernels of a side, and the G1 and G2 sides of a proof,
  // are independent until their reductions; on small batches none of them fills the chip on its own.
  hipStream_t side[3] = {nullptr, nullptr, nullptr};
  hipEvent_t sev[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  hipStream_t cur = nullptr;  // launch target override (nullptr = the context's stream)
  bool overlap = true;
  // SIMD slots of the device (CUs x 4); every heavy kernel runs one 512-VGPR wave per SIMD
  size_t simd_slots = 1024;
  // planner overrides (gs_set_option; 0 / -1 = planned per batch)
  int var_tm = 0;
  int red_k = 0;  // outputs per reduction lane (1, 2, 4); 0 = planned
  int var_ws_lanes = 0;  // Straus lanes per launch (their table workspace is 3.5 .. 28 KB each); 0 = 2^19
  int var_mo = 0, var_w = 0;  // outputs per Straus lane (1, 2, 4) and its window width (4, 5); 0 = planned
  int miller_ch = 0, miller_twin = -1;
  int coop_fe = 1;  // 0 never, 1 when one lane per final exponentiation cannot fill the chip, 2 always
  bool line_tables = true;  // pairs whose G2 argument is a CRS element read precomputed Miller lines
  // host-pointer entry points (HostPipe below): pinned staging, a copy stream, memcpy workers, per-array events
  struct HostPipe* pipe = nullptr;  // set while a host-pointer call is enqueuing its kernels
  struct CopyPool* pool = nullptr;
  hipStream_t copy_stream = nullptr;
  void* pin = nullptr;
  size_t pin_cap = 0;
  hipEvent_t pev[16] = {nullptr};
  // mixed calls: launches are recorded per part and merged (launch_seg / replay); scratch buffers get a per-part tag;
  // the lane-shape planners see the batch size the merged launches will have
  struct Recorder* rec = nullptr;
  // 1: GLV / psi-GLS scalar multiplications (r-torsion points only, as arkworks' deserialisation guarantees); 0: every
  // variable-base scalar multiplication is a plain signed-window double-and-add lane (k_var.plain): ANY curve point,
  // like the reference's Com::scalar_mul (data_structures.rs:336-342), ~2.5x the variable-base work
  bool endo = true;
  int var_w2 = -1;  // G1 Straus lanes in the kernels built for two waves per SIMD: -1 planned, 0 never, 1 always
  int var_tab = -1;      // verifier's Gamma^T c on shared per-base window tables: -1 planned (large arities), 0, 1
  int mixed_merge = -1;  // -1 planned (merge while the parts cannot fill the chip on their own), 0 never, 1 always
  int scratch_tag = 0;
  size_t fill_n = 0;
  hipStream_t copy_stream2 = nullptr, copy_out_stream = nullptr, copy_out_stream2 = nullptr;  // two per direction
  hipEvent_t pev2[16] = {nullptr};
  hipEvent_t pev_ready = nullptr;
  // buffers a regrow or a plan eviction would have freed WHILE a mixed call records its launches (c->rec): the recorded
  // argument packs hold raw device pointers, so nothing is freed until the replay has been enqueued and drained
  std::vector<void*> deferred_free;
  unsigned long long* stamp = nullptr;  // 3 x u64 on the device: the clock stamps of the launch being profiled
};

// Every live context of the process (gs_ctx_create .. gs_ctx_destroy): page-locked caller ranges are process-wide
// (g_reg below), so releasing one has to drain the copy queues of EVERY context that may be moving it -- the shards of
// a gs_multi context are contexts of their own.
static std::mutex g_ctx_mu;
static std::vector<gs_ctx*> g_ctx_all;
// all DMA and kernels of one context done (its compute stream, side streams and the four copy streams)
static void drain_ctx(gs_ctx* c) {
  hipStreamSynchronize(c->stream);
  for (hipStream_t st : c->side)
    if (st) hipStreamSynchronize(st);
  for (hipStream_t st : {c->copy_stream, c->copy_stream2, c->copy_out_stream, c->copy_out_stream2})
    if (st) hipStreamSynchronize(st);
}

static int fail(gs_ctx* c, int code, const char* what, hipError_t e = hipSuccess) {
  if (c) {
    c->err = what;
    if (e != hipSuccess) {
      c->err += ": ";
      c->err += hipGetErrorString(e);
    }
  }
  return code;
}
#define HIPCHK(ctx, call)                                          \
  do {                                                             \
    hipError_t e_ = (call);                                        \
    if (e_ != hipSuccess) return fail(ctx, GS_ERR_DEVICE, #call, e_); \
  } while (0)

#define RC(...)                   \
  do {                            \
    int rc_ = (__VA_ARGS__);      \
    if (rc_ != GS_OK) return rc_; \
  } while (0)

static int ensure(gs_ctx* c, DevBuf& b, size_t bytes) {
  if (bytes <= b.cap && b.p) return GS_OK;
  if (b.p) {  // growing: kernels already enqueued may still read the old buffer
    if (c && c->rec) {
      // recording: nothing has been launched, a sync protects nothing -- the launches that will read the old buffer
      // are in the recorder.  Park it; mixed_run / mixed_host release the list after the replay.
      c->deferred_free.push_back(b.p);
    } else {
      if (c) {
        hipStreamSynchronize(c->stream);
        for (hipStream_t st : c->side)
          if (st) hipStreamSynchronize(st);
      }
      hipFree(b.p);
    }
  }
  b.p = nullptr;
  b.cap = 0;
  size_t want = bytes < 256 ? 256 : bytes;
  hipError_t e = hipMalloc(&b.p, want);
  if (e != hipSuccess) return fail(c, GS_ERR_ALLOC, "hipMalloc", e);
  b.cap = want;
  return GS_OK;
}
static int scratch(gs_ctx* c, const char* name, size_t bytes, void** out) {
  char tagged[96];
  if (c->scratch_tag) {  // a part of a mixed call: its buffers are live next to the other parts'
    snprintf(tagged, sizeof tagged, "m%d:%s", c->scratch_tag, name);
    name = tagged;
  }
  DevBuf& b = c->scratch[name];
  int rc = ensure(c, b, bytes);
  *out = b.p;
  return rc;
}

// ---- external-profiler markers ("tracing") ---------------------------------------------
// roctxRangePush / Pop around the phases of prove and verify so that `rocprofv3 --marker-trace` segments a step.
// librocprofiler-sdk-roctx (or the older libroctx64) is bound at run time and only if present; without it the ranges
// are no-ops.
struct Roctx {
  int (*push)(const char*) = nullptr;
  int (*pop)() = nullptr;
  Roctx() {
    const char* names[] = {"librocprofiler-sdk-roctx.so.1", "librocprofiler-sdk-roctx.so", "libroctx64.so.4", "libroctx64.so"};
    void* h = nullptr;
    for (const char* n : names)
      if ((h = dlopen(n, RTLD_NOW | RTLD_NOLOAD | RTLD_LOCAL))) break;
    if (!h && getenv("GS_ROCTX"))  // not mapped by a profiler: load it only when asked to
      for (const char* n : names)
        if ((h = dlopen(n, RTLD_NOW | RTLD_LOCAL))) break;
    if (!h) return;
    push = (int (*)(const char*))dlsym(h, "roctxRangePushA");
    pop = (int (*)())dlsym(h, "roctxRangePop");
    if (!push || !pop) push = nullptr, pop = nullptr;
  }
};
static Roctx& roctx() {
  static Roctx r;
  return r;
}
struct RangeGuard {
  bool on;
  explicit RangeGuard(const char* name) : on(roctx().push != nullptr) {
    if (on) roctx().push(name);
  }
  ~RangeGuard() {
    if (on) roctx().pop();
  }
};

// ---- host-pointer pipeline ---------------------------------------------------------------------------------------
// What a drop-in caller hands over are host slices (prove.rs:29-52, verifier.rs:18-21).  Pageable hipMemcpy on the
// compute stream cost 15 % of a 2^16 step (profiles/r2/host_path_rate.txt).  The un-suffixed entry points now
//   1. copy every input array into a grow-only PINNED staging buffer with a few memcpy worker threads (CopyPool),
//      arrays in the order the kernels need them;
//   2. enqueue the H2D of an array on a separate copy stream as soon as its staging copy is complete, with one event
//      per array;
//   3. let the engine wait for exactly the arrays the next kernels read (need(): scalars before the preparation
//      kernel, the G1 arguments before the G1 side, the G2 arguments before the G2 side / the Miller loop), so that the
//      upload of the later arrays runs under the kernels of the earlier ones;
//   4. bring the outputs back array by array (D2H into pinned memory, then the workers copy into the caller's
//      buffers while the next array is in flight).
// Equation-chunking the kernels instead would cost more than it hides: the lane shapes want whole batches (2^12
// equations run at 0.6 of the 2^16 rate).
struct CopyPool {
  struct Job {
    void* dst;
    const void* src;
    size_t n;
    std::atomic<int>* left;
  };
  std::vector<std::thread> th;
  std::mutex mu;
  std::condition_variable cv;
  std::deque<Job> q;
  bool stop = false;
  explicit CopyPool(int nthreads) {
    for (int i = 0; i < nthreads; i++)
      th.emplace_back([this] {
        for (;;) {
          Job j;
          {
            std::unique_lock<std::mutex> lk(mu);
            cv.wait(lk, [this] { return stop || !q.empty(); });
            if (q.empty()) return;
            j = q.front();
            q.pop_front();
          }
          memcpy(j.dst, j.src, j.n);
          j.left->fetch_sub(1, std::memory_order_release);
        }
      });
  }
  ~CopyPool() {
    {
      std::lock_guard<std::mutex> lk(mu);
      stop = true;
    }
    cv.notify_all();
    for (auto& t : th) t.join();
  }
  // copy [src, src + n) to dst in pieces; *left counts the pieces still to do
  void submit(void* dst, const void* src, size_t n, std::atomic<int>* left) {
    // 2 MB pieces for large arrays; a mid-size array (the 1.5-6 MB arrays of a 2^12 batch) is cut finer so that every
    // worker gets a share of it
    size_t piece = n / (2 * th.size() + 1);
    piece = std::min(std::max(piece, (size_t)256 << 10), (size_t)2 << 20) & ~(size_t)4095;
    size_t np = (n + piece - 1) / piece;
    left->fetch_add((int)np, std::memory_order_relaxed);
    {
      std::lock_guard<std::mutex> lk(mu);
      for (size_t i = 0; i < np; i++) {
        size_t o = i * piece;
        q.push_back(Job{(uint8_t*)dst + o, (const uint8_t*)src + o, std::min(piece, n - o), left});
      }
    }
    cv.notify_all();
  }
  static void wait(std::atomic<int>* left) {
    while (left->load(std::memory_order_acquire) > 0) std::this_thread::yield();
  }
};

static std::mutex g_reg_mu;
static std::map<const uint8_t*, size_t> g_reg;  // gs_host_register: start -> bytes
struct HostPipe {
  struct Arr {
    const void* hin = nullptr;
    void* hout = nullptr;
    void* d = nullptr;
    size_t bytes = 0, off = 0;
    std::atomic<int> left{0};
    bool enq = false;
    bool direct = false;  // the caller's array is page-locked already: DMA to / from it, no staging copy
  };
  // Page-locked caller memory is what was registered through gs_host_register -- the library's own list, nothing else.
  // (The first version asked the runtime, hipPointerGetAttributes on both ends of the array.  That also reports ranges
  // the runtime itself has pinned behind the caller's back -- the source of an earlier pageable hipMemcpy, kept in its
  // pin cache, possibly mapped READ-ONLY, possibly belonging to a buffer that has been freed since and whose addresses a
  // new array now occupies.  A D2H straight into such a range ended in "write access to a read-only page" on the GPU.)
  static bool host_pinned(const void* p, size_t bytes);
  gs_ctx* c;
  Arr arr[16];
  hipEvent_t ev1[16] = {nullptr}, ev2[16] = {nullptr};
  int out_order[16], n_out = 0;  // output arrays in the order their D2H went out (= the order they arrive)
  int n = 0;
  bool trace = false;
  bool begun = false, finished = false;  // begin() has enqueued work on the copy streams / finish() has waited for all of it
  int split = 2;  // DMA transfers per array and direction (two copy streams: two SDMA engines)
  std::chrono::steady_clock::time_point t0;
  explicit HostPipe(gs_ctx* ctx) : c(ctx) {}
  ~HostPipe() {
    for (int i = 0; i < n; i++) CopyPool::wait(&arr[i].left);  // no worker may still touch the caller's memory
    // A call that failed between begin() and the end of finish() returns with H2D / D2H still queued against the caller's
    // (possibly page-locked) arrays and against stage.* buffers the next call will overwrite: nothing may be in flight
    // when the caller gets the error back and unwinds (unregisters, frees).
    if (begun && !finished) drain_ctx(c);
    for (int i = 0; i < 16; i++)
      for (hipEvent_t ev : {ev1[i], ev2[i]})
        if (ev) hipEventDestroy(ev);
    if (c->pipe == this) c->pipe = nullptr;
  }
  double ms() const { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); }
  // declare array `i` (slots are fixed per entry point: the need() masks name them); null / empty arrays stay absent
  int in(int i, const void* h, size_t bytes) {
    if (h && bytes) arr[i].hin = h, arr[i].bytes = bytes;
    n = std::max(n, i + 1);
    return GS_OK;
  }
  int out(int i, void* h, size_t bytes) {
    if (h && bytes) arr[i].hout = h, arr[i].bytes = bytes;
    n = std::max(n, i + 1);
    return GS_OK;
  }
  void* dev(int i) const { return arr[i].d; }
  // bytes of pinned staging this call needs (every declared array, 256-byte aligned)
  size_t layout() {
    size_t total = 0;
    for (int i = 0; i < n; i++) {
      if (!arr[i].bytes) continue;
      arr[i].direct = host_pinned(arr[i].hin ? arr[i].hin : arr[i].hout, arr[i].bytes);
      if (arr[i].direct) continue;
      arr[i].off = total;
      total += (arr[i].bytes + 255) & ~(size_t)255;
    }
    return total;
  }
  // the context's grow-only pinned buffer holds at least `bytes` (never call between begin() and finish() of a pipe
  // that uses it: growing moves it)
  static int pin_reserve(gs_ctx* c, size_t bytes) {
    if (bytes <= c->pin_cap) return GS_OK;
    if (c->pin) hipHostFree(c->pin);
    c->pin = nullptr;
    c->pin_cap = 0;
    hipError_t e = hipHostMalloc(&c->pin, bytes, hipHostMallocDefault);
    if (e != hipSuccess) return fail(c, GS_ERR_ALLOC, "hipHostMalloc (pinned staging)", e);
    c->pin_cap = bytes;
    return GS_OK;
  }
  // device slots, workers; then the staging copies start, inputs in the order `order` lists them.  `base`: where this
  // pipe's region of the pinned buffer begins (several pipes of one mixed call share the buffer)
  int begin(const int* order, int norder, size_t base = 0, bool reserve = true) {
    begun = true;
    t0 = std::chrono::steady_clock::now();
    trace = getenv("GS_PIPE_TRACE") != nullptr;
    if (const char* e = getenv("GS_COPY_SPLIT")) split = std::max(1, std::min(2, atoi(e)));
    // The copy streams are created at HIGH priority, which is not about urgency: the runtime multiplexes the streams
    // of one priority level over a few hardware queues (4 by default), and a copy stream that lands on the compute
    // stream's queue is serialized with the kernels there -- the first kernel then starts only after EVERY upload
    // enqueued before it, and uploads enqueued behind a long kernel wait for that kernel
    // (profiles/r3/host_pipe_timeline.txt).  Another priority level is another set of queues.
    int prio_least = 0, prio_greatest = 0;
    (void)hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest);
    for (hipStream_t* st : {&c->copy_stream, &c->copy_stream2, &c->copy_out_stream, &c->copy_out_stream2})
      if (!*st && hipStreamCreateWithPriority(st, hipStreamNonBlocking, prio_greatest) != hipSuccess) {
        (void)hipGetLastError();  // no priority levels here: an ordinary stream (correct, possibly serialized as above)
        *st = nullptr;
        HIPCHK(c, hipStreamCreateWithFlags(st, hipStreamNonBlocking));
      }
    if (!c->pev_ready) HIPCHK(c, hipEventCreateWithFlags(&c->pev_ready, hipEventDisableTiming));
    if (!c->pool) {
      int nt = 4;
      if (const char* e = getenv("GS_COPY_THREADS")) nt = std::max(1, std::min(32, atoi(e)));
      c->pool = new CopyPool(nt);
    }
    const size_t total = layout();
    if (reserve) RC(pin_reserve(c, base + total));
    for (int i = 0; i < n; i++) {
      if (!arr[i].bytes) continue;
      char name[32];
      snprintf(name, sizeof name, "stage.%d", i);
      RC(scratch(c, name, arr[i].bytes, &arr[i].d));
      arr[i].off += base;
      // events are per (pipe, array): a mixed call has several pipes in flight
      for (hipEvent_t* ev : {&ev1[i], &ev2[i]}) HIPCHK(c, hipEventCreateWithFlags(ev, hipEventDisableTiming));
    }
    for (int k = 0; k < norder; k++) {
      Arr& a = arr[order[k]];
      if (a.hin && !a.direct) c->pool->submit((uint8_t*)c->pin + a.off, a.hin, a.bytes, &a.left);
    }
    for (int k = 0; k < norder; k++) {  // page-locked inputs need no staging: their uploads go out at once
      Arr& a = arr[order[k]];
      if (!a.hin || !a.direct || a.enq) continue;
      RC(xfer(a.d, a.hin, a.bytes, hipMemcpyHostToDevice, c->copy_stream, c->copy_stream2, ev1[order[k]], ev2[order[k]]));
      a.enq = true;
    }
    c->pipe = this;
    if (trace) fprintf(stderr, "[pipe] %7.2f ms begin done (%zu bytes of staging)\n", ms(), total);
    return GS_OK;
  }
  // (begun is set at the top of begin(): a begin() that fails half way has already queued copies)
  // one array across PCIe, in `split` pieces on as many copy streams; ev is recorded (on s0) behind all of them
  int xfer(void* dst, const void* src, size_t bytes, hipMemcpyKind kind, hipStream_t s0, hipStream_t s1, hipEvent_t ev,
           hipEvent_t ev_b) {
    if (split < 2 || bytes < ((size_t)4 << 20)) {
      HIPCHK(c, hipMemcpyAsync(dst, src, bytes, kind, s0));
    } else {
      size_t h = (bytes / 2 + 255) & ~(size_t)255;
      HIPCHK(c, hipMemcpyAsync(dst, src, h, kind, s0));
      HIPCHK(c, hipMemcpyAsync((uint8_t*)dst + h, (const uint8_t*)src + h, bytes - h, kind, s1));
      HIPCHK(c, hipEventRecord(ev_b, s1));
      HIPCHK(c, hipStreamWaitEvent(s0, ev_b, 0));
    }
    HIPCHK(c, hipEventRecord(ev, s0));
    return GS_OK;
  }
  // the kernels about to be enqueued read the input arrays in `mask`: their uploads go out (if they have not yet)
  // and the target stream waits for them
  int need(unsigned mask) {
    hipStream_t tgt = c->cur ? c->cur : c->stream;
    for (int i = 0; i < n; i++) {
      Arr& a = arr[i];
      if (!((mask >> i) & 1) || !a.hin) continue;
      if (!a.enq) {
        double w0 = trace ? ms() : 0;
        CopyPool::wait(&a.left);
        double w1 = trace ? ms() : 0;
        RC(xfer(a.d, a.direct ? a.hin : (uint8_t*)c->pin + a.off, a.bytes, hipMemcpyHostToDevice, c->copy_stream, c->copy_stream2, ev1[i],
                ev2[i]));
        a.enq = true;
        if (trace)
          fprintf(stderr, "[pipe] %7.2f ms H2D of array %d enqueued (%.1f MB, waited %.2f ms for its staging, %.2f ms in the enqueue)\n",
                  ms(), i, a.bytes / 1e6, w1 - w0, ms() - w1);
      }
      HIPCHK(c, hipStreamWaitEvent(tgt, ev1[i], 0));
    }
    return GS_OK;
  }
  // the output arrays in `mask` are complete once the work enqueued so far on the current stream is done: their D2H
  // starts behind it on the copy-out streams, under whatever kernels follow
  int out_ready(unsigned mask) {
    if (c->rec) return GS_OK;  // launches are only being recorded: nothing has been computed yet
    hipStream_t src = c->cur ? c->cur : c->stream;
    bool first = true;
    for (int i = 0; i < n; i++) {
      Arr& a = arr[i];
      if (!((mask >> i) & 1) || !a.hout || a.enq) continue;
      if (first) {
        HIPCHK(c, hipEventRecord(c->pev_ready, src));
        HIPCHK(c, hipStreamWaitEvent(c->copy_out_stream, c->pev_ready, 0));
        HIPCHK(c, hipStreamWaitEvent(c->copy_out_stream2, c->pev_ready, 0));
        first = false;
      }
      RC(xfer(a.direct ? a.hout : (uint8_t*)c->pin + a.off, a.d, a.bytes, hipMemcpyDeviceToHost, c->copy_out_stream, c->copy_out_stream2,
              ev1[i], ev2[i]));
      a.enq = true;
      out_order[n_out++] = i;
      if (trace) fprintf(stderr, "[pipe] %7.2f ms D2H of array %d enqueued (%.1f MB)\n", ms(), i, a.bytes / 1e6);
    }
    return GS_OK;
  }
  // outputs: D2H per array behind the kernels, each copied on to the caller's buffer while the next one is in flight
  int finish() {
    c->pipe = nullptr;
    if (trace) fprintf(stderr, "[pipe] %7.2f ms kernels enqueued\n", ms());
    hipStream_t keep = c->cur;
    c->cur = nullptr;
    int rc = out_ready(0xFFFFu);  // whatever has not gone out yet, behind the last kernel
    c->cur = keep;
    RC(rc);
    for (int k = 0; k < n_out; k++) {  // in arrival order: an early array is on its way to the caller while later ones fly
      const int i = out_order[k];
      Arr& a = arr[i];
      HIPCHK(c, hipEventSynchronize(ev1[i]));
      if (trace) fprintf(stderr, "[pipe] %7.2f ms array %d arrived\n", ms(), i);
      if (!a.direct) c->pool->submit(a.hout, (uint8_t*)c->pin + a.off, a.bytes, &a.left);
    }
    HIPCHK(c, hipStreamSynchronize(c->stream));
    for (int i = 0; i < n; i++) CopyPool::wait(&arr[i].left);
    // (every D2H was waited for through its event above; the H2D queues were waited for by the kernels that the
    // stream sync has just seen finish)
    finished = true;
    if (trace) fprintf(stderr, "[pipe] %7.2f ms done\n", ms());
    return GS_OK;
  }
};
bool HostPipe::host_pinned(const void* p, size_t bytes) {
  static const bool off = getenv("GS_PIPE_NO_DIRECT") != nullptr;
  if (off || !p || !bytes) return false;
  std::lock_guard<std::mutex> lk(g_reg_mu);
  auto it = g_reg.upper_bound((const uint8_t*)p);
  if (it == g_reg.begin()) return false;
  --it;  // the last registration that starts at or before p
  return (const uint8_t*)p + bytes <= it->first + it->second;
}
static inline int need(gs_ctx* c, unsigned mask) { return c->pipe ? c->pipe->need(mask) : GS_OK; }
static inline int out_ready(gs_ctx* c, unsigned mask) { return c->pipe ? c->pipe->out_ready(mask) : GS_OK; }
// array slots of the host-pointer entry points
enum { PI_X = 0, PI_Y, PI_A, PI_B, PI_G, PI_R, PI_S, PI_T, PO_XC, PO_YC, PO_PI, PO_TH };           // prove
enum { VI_A = 0, VI_B, VI_G, VI_TG, VI_XC, VI_YC, VI_PI, VI_TH, VO_OK };                            // verify
#define BIT(i) (1u << (i))

// the batch size the lane-shape planners should fill the chip for: the part's own, or (mixed calls, whose parts'
// launches are merged) the whole call's
static inline size_t fillN(const gs_ctx* c, size_t N) { return c->fill_n > N ? c->fill_n : N; }

// ---- Miller-lane planning --------------------------------------------------------------------------
// A Miller lane carries up to MILLER_CH (P, Q) pairs (twin: (Q, P0, P1) triples with two accumulators) and squares its
// own accumulator every iteration, so longer lanes do less total work but a small batch needs many short ones to fill
// the chip.  Pairs whose G2 argument is a CRS element read precomputed lines and are cheaper.  Cost model (checked
// against measurements at 2^12..2^16): rounds of waves over the SIMD slots x the longest lane, in
// Fq multiplications (profiles/r1/fq_mul_counts.json).  Lanes are task-major (a wave = 64 equations of ONE task).
struct MCost {
  double base, var, fix;
};
// whether the planner may pick the lane-pair form of the twin loop on its own (measured per round: DESIGN.md section 4.2)
#ifndef GS_PLAN_PAIR
#define GS_PLAN_PAIR 1
#endif
// (profiles/r3/fq_mul_counts.json: miller*_per_lane, _per_pair / _per_triple, _per_fixed_pair / _per_fixed_triple --
// the counts of the unpaired line products both curves use since round 3)
static inline MCost mcost(int curve, bool twin) {
  if (curve == 0) return twin ? MCost{4536.0, 7764.0, 6256.0} : MCost{2268.0, 4636.0, 3128.0};
  return twin ? MCost{4680.0, 10519.0, 8096.0} : MCost{2340.0, 6471.0, 4048.0};
}
static inline bool pair_fixed(bool lt, const PairRef& r) { return lt && r.q_arr == 2; }  // Q array 2 = CRS (v, W2)
// Split one cell's pairs into the fewest tasks whose lane cost stays within `budget`: costly pairs first, each to the
// lightest task that still has room (LPT).
static void chunk_tasks(std::vector<MillerTask>& mt, const std::vector<PairRef>& pr, double budget, int b, bool single,
                        const MCost& mc, bool lt) {
  size_t P = pr.size();
  if (P == 0) return;
  std::vector<size_t> order;
  double total = 0;
  for (size_t i = 0; i < P; i++)
    if (!pair_fixed(lt, pr[i])) order.push_back(i);
  for (size_t i = 0; i < P; i++)
    if (pair_fixed(lt, pr[i])) order.push_back(i);
  for (size_t i = 0; i < P; i++) total += pair_fixed(lt, pr[i]) ? mc.fix : mc.var;
  // no fewer tasks than the capacity or the budget allow (large arities have hundreds of pairs per cell: start at the
  // bound instead of walking up to it, and place with a heap: O(P log P) per attempt)
  size_t nt0 = (P + MILLER_CH - 1) / MILLER_CH;
  if (budget > 0) nt0 = std::max(nt0, (size_t)(total / budget));
  nt0 = std::min(std::max(nt0, (size_t)1), P);
  for (size_t nt = nt0; nt <= P; nt++) {
    std::vector<double> load(nt, 0.0);
    std::vector<std::vector<size_t>> members(nt);
    typedef std::pair<double, size_t> Slot;  // (load, task): lightest task that still has room first
    std::priority_queue<Slot, std::vector<Slot>, std::greater<Slot>> heap;
    for (size_t t = 0; t < nt; t++) heap.push(Slot(0.0, t));
    for (size_t idx : order) {
      Slot sl = heap.top();
      heap.pop();
      size_t best = sl.second;
      members[best].push_back(idx);
      load[best] += pair_fixed(lt, pr[idx]) ? mc.fix : mc.var;
      if (members[best].size() < (size_t)MILLER_CH) heap.push(Slot(load[best], best));
    }
    double mx = 0;
    for (double l : load) mx = l > mx ? l : mx;
    if (mx > budget && nt < P) continue;
    for (size_t t = 0; t < nt; t++) {
      MillerTask mtk;
      memset(&mtk, 0, sizeof mtk);
      mtk.b = (uint8_t)b;
      mtk.single = single ? 1 : 0;
      mtk.np = (uint8_t)members[t].size();
      for (int q = 0; q < mtk.np; q++) mtk.pr[q] = pr[members[t][q]];
      mt.push_back(mtk);
    }
    return;
  }
}
// serial cost (in units of `unit`) of reducing `n` partial results: runs of 8 folded in parallel while more than 16
// are left (k_cell_fold / k_slot_fold), the rest by the consuming lane
static double fold_cost(double n, double unit) {
  double c = 0;
  while (n > 16.0) {
    c += 8.0 * unit;
    n = (double)(size_t)((n + 7.0) / 8.0);
  }
  return c + n * unit;
}
// `pair`: the twin task list run by k_miller_pair -- two lanes of half the length per (equation, task)
// A pair of lanes (k_miller_pair) is one task inside ONE wave: what the wave executes per loop digit is a squaring, one
// twist-point step per ROUND of two stepping triples (the lanes of a pair step one each; an odd triple is a round of its
// own) and one line product per triple.
static inline double pair_lane_cost(const gs_ctx* c, const MillerTask& t) {
  const MCost mc = mcost(c->curve, true);
  int nv = 0, nf = 0;
  for (int q = 0; q < t.np; q++) (pair_fixed(c->line_tables, t.pr[q]) ? nf : nv)++;
  return mc.base / 2 + ((nv + 1) / 2) * (mc.var - mc.fix) + (nv + nf) * mc.fix / 2;
}
static double miller_cost(const gs_ctx* c, size_t N, const std::vector<MillerTask>& mt, bool twin, bool pair = false) {
  MCost mc = mcost(c->curve, twin);
  // Lanes are task-major: a wave is 64 equations of ONE task and lasts as long as that task's lane.  A launch lasts
  // as long as its longest lane, or -- once it is several rounds of waves -- as long as all lanes together take on the
  // SIMD slots (waves of short tasks do not wait for those of long ones).
  double longest = 0, all = 0;
  // waves per task (below 64 equations a wave holds several tasks: a fraction of a wave each)
  const size_t NL = pair ? 2 * N : N;  // lanes per task
  const double wpt = NL >= 64 ? (double)(size_t)((NL + 63) / 64) : (double)NL / 64.0;
  for (const MillerTask& t : mt) {
    double l = mc.base;
    for (int q = 0; q < t.np; q++) l += pair_fixed(c->line_tables, t.pr[q]) ? mc.fix : mc.var;
    if (pair) l = pair_lane_cost(c, t);
    longest = l > longest ? l : longest;
    all += l * wpt;
  }
  double rounds = (double)mt.size() * wpt / (double)c->simd_slots;
  double fill = all / (double)c->simd_slots;
  if (rounds > 1.0 && rounds < 4.0) fill *= (double)(size_t)(rounds + 0.999) / rounds;  // whole rounds while few
  double lane = rounds <= 1.0 ? longest : (fill > longest ? fill : longest);
  // + the lanes that multiply a cell's partials together (54 Fq multiplications each): k_final's own serial product
  // up to 16 of them, above that K-ary tree levels of 8 (k_cell_fold; large arities)
  double per_cell = (double)mt.size() * (twin ? 2.0 : 1.0) / 4.0;
  return lane + fold_cost(per_cell, 54.0);
}
// lane-cost budgets worth trying: a variable + f fixed pairs
static std::vector<double> miller_budgets(const gs_ctx* c, bool twin) {
  const MCost mc = mcost(c->curve, twin);
  std::vector<double> out;
  for (int a = 1; a <= MILLER_CH; a++)
    for (int f = 0; f <= 2 && a + f <= MILLER_CH; f++) {
      if (c->miller_ch > 0 && a + f != c->miller_ch) continue;
      out.push_back(a * mc.var + f * mc.fix);
    }
  return out;
}

// k_var_multi lanes per launch: their Straus tables (affine table + the Jacobian staging of its build) live in a
// per-context workspace of 9 .. 70 KB per lane (G2, 8 terms, 5-bit windows: 28 + 42 KB).  Two rounds of resident waves
// (simd_slots x 64 lanes each) keep the launch overhead below 1 % and bound the workspace at 9.2 GB for the largest
// lane whatever the batch.
static inline size_t var_ws_default(const gs_ctx* c) { return 2 * c->simd_slots * 64; }
// Straus lanes per launch.  Two rounds of resident waves is what a 2^16 batch needs; a larger batch used to run as several
// launches over that workspace, and every launch boundary idles the SIMDs whose last wave finished early (2^17 / 2^18 on
// one GPU ran 5 % below the 2^16 rate: profiles/r3/bench_log2n17/18.json).  This GPU has 288 GB: when a quarter of
// the free device memory (at most 32 GB) holds the workspace of more lanes, the launch takes them -- whole rounds of
// waves, so that a remaining chunk is never a sliver.
static size_t var_ws_auto(const gs_ctx* c, size_t tot, size_t bytes_per_lane) {
  const size_t round = c->simd_slots * 64, base = var_ws_default(c);
  if (tot <= base) return tot;
  size_t free_b = 0, total_b = 0;
  if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) {
    (void)hipGetLastError();
    return base;
  }
  size_t budget = std::min(free_b / 4, (size_t)32 << 30);
  size_t lanes = budget / std::max(bytes_per_lane, (size_t)1) / round * round;
  return std::min(tot, std::max(lanes, base));
}
// launch wrapper with optional HIP-event timing (used by bench.py's roofline leg)
template <class K, class... Args>
static int launch(gs_ctx* c, const char* name, K kern, size_t total, int block, Args... args) {
  if (c->rec) return fail(c, GS_ERR_ARG, "internal: this kernel cannot be part of a merged launch");
  if (total == 0) return GS_OK;
  unsigned grid = (unsigned)((total + block - 1) / block);
  hipStream_t st = c->cur ? c->cur : c->stream;
  if (c->prof) hipEventRecord(c->ev0, st);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(block), 0, st, args...);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(c, GS_ERR_DEVICE, name, e);
  if (c->prof) {
    hipEventRecord(c->ev1, st);
    hipEventSynchronize(c->ev1);
    float ms = 0;
    hipEventElapsedTime(&ms, c->ev0, c->ev1);
    if (!c->prof_map.count(name)) c->prof_order.push_back(name);
    ProfEntry& p = c->prof_map[name];
    p.ms += ms;
    p.n += 1;
    p.lanes += total;
    p.work += c->work_hint ? c->work_hint : total;
  }
  c->work_hint = 0;
  return GS_OK;
}

// ---- segmented launches (csrc/gs_kernels.cuh: k_seg) ----------------------------------------------------------------
// launch_seg<Body>(c, name, lanes, block, args...) runs a kernel BODY over one segment -- or, while a mixed call is
// recording (c->rec), only notes the launch.  replay() then walks the parts' launch lists in step and merges the launches
// that run the same body (same instantiation, same block size) into one k_seg launch with a segment per part.
struct LaunchRec {
  std::string name;
  const void* kern = nullptr;  // address of the k_seg instantiation: the identity of the body
  size_t total = 0;
  int block = 64;
  uint64_t work = 0;
  std::vector<uint8_t> pack;   // the bytes of this launch's Pack<args...>
  int (*go)(gs_ctx*, const LaunchRec* const*, int) = nullptr;
};
struct Recorder {
  std::vector<std::vector<LaunchRec>> parts;
  int cur = 0;
};
template <class Body, class... A> static int go_seg(gs_ctx* c, const LaunchRec* const* r, int nr) {
  Segs<A...> S;
  memset((void*)&S, 0, sizeof S);
  S.n = nr;
  size_t lanes = 0, tot = 0;
  uint64_t work = 0;
  const int block = r[0]->block;
  for (int i = 0; i < nr; i++) {
    S.lo[i] = lanes;
    memcpy((void*)&S.a[i], r[i]->pack.data(), sizeof(Pack<A...>));
    lanes += (r[i]->total + block - 1) / block * block;
    tot += r[i]->total;
    work += r[i]->work ? r[i]->work : r[i]->total;
  }
  if (lanes == 0) return GS_OK;
  hipStream_t st = c->cur ? c->cur : c->stream;
  if (c->prof) {
    if (!c->stamp && hipMalloc((void**)&c->stamp, 3 * sizeof(unsigned long long)) != hipSuccess) c->stamp = nullptr;
    if (c->stamp) hipMemsetAsync(c->stamp, 0, 3 * sizeof(unsigned long long), st);
    S.stamp = c->stamp;
    hipEventRecord(c->ev0, st);
  }
  hipLaunchKernelGGL((k_seg<Body, A...>), dim3((unsigned)(lanes / block)), dim3(block), 0, st, S);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(c, GS_ERR_DEVICE, r[0]->name.c_str(), e);
  if (c->prof) {
    hipEventRecord(c->ev1, st);
    hipEventSynchronize(c->ev1);
    float ms = 0;
    hipEventElapsedTime(&ms, c->ev0, c->ev1);
    const std::string& name = r[0]->name;
    if (!c->prof_map.count(name)) c->prof_order.push_back(name);
    ProfEntry& p = c->prof_map[name];
    p.ms += ms;
    p.n += 1;
    p.lanes += tot;
    p.work += work;
    if (c->stamp) {
      unsigned long long h[3] = {0, 0, 0};
      if (hipMemcpy(h, c->stamp, sizeof h, hipMemcpyDeviceToHost) == hipSuccess) p.cyc += (double)h[0], p.rt += (double)h[1];
    }
  }
  return GS_OK;
}
template <class Body, class... A, class... U>
static int launch_seg_typed(gs_ctx* c, const char* name, void (*)(size_t, A...), size_t total, int block, U&&... u) {
  static_assert(sizeof...(A) == sizeof...(U), "argument count of the kernel body");
  static_assert(sizeof(Segs<A...>) <= 3800, "the segment table travels as a kernel argument");
  LaunchRec r;
  r.name = name;
  r.kern = (const void*)&k_seg<Body, A...>;
  r.total = total;
  r.block = block;
  r.work = c->work_hint;
  c->work_hint = 0;
  Pack<A...> pk{{static_cast<A>(u)}...};
  r.pack.assign((const uint8_t*)&pk, (const uint8_t*)&pk + sizeof pk);
  r.go = &go_seg<Body, A...>;
  if (c->rec) {
    if (total) c->rec->parts[c->rec->cur].push_back(std::move(r));
    return GS_OK;
  }
  if (total == 0) return GS_OK;
  const LaunchRec* rp = &r;
  return r.go(c, &rp, 1);
}
template <class Body, class... U> static int launch_seg(gs_ctx* c, const char* name, size_t total, int block, U&&... u) {
  typedef decltype(&Body::run) Sig;  // void (*)(size_t g, args...): the body's own parameter types
  return launch_seg_typed<Body>(c, name, (Sig) nullptr, total, block, std::forward<U>(u)...);
}
// Launch what the parts of a mixed call recorded.  Each part's list is in dependency order (it is what the part would
// have enqueued on the stream); launches of different parts are independent.  Lock-step merge: the parts whose NEXT
// launch runs the same body (same instantiation, same block size) form a group that goes out as one segmented launch.
// When the heads differ (one part has a kernel the others lack, or another variant of it), a group is taken that no
// other part could still join later -- its body does not occur further down any non-member's list -- so that the parts
// fall back into step right after it; only if every head is awaited elsewhere does the first unfinished part lead.
static int replay(gs_ctx* c, Recorder& R) {
  const size_t np = R.parts.size();
  std::vector<size_t> cur(np, 0);
  auto head = [&](size_t p) -> const LaunchRec* { return cur[p] < R.parts[p].size() ? &R.parts[p][cur[p]] : nullptr; };
  auto same = [](const LaunchRec* a, const LaunchRec* b) { return a->kern == b->kern && a->block == b->block; };
  for (;;) {
    size_t lead = np;
    for (size_t p = 0; p < np && lead == np; p++) {
      const LaunchRec* h = head(p);
      if (!h) continue;
      bool awaited = false;  // does a part that is NOT at this body now still have it ahead?
      for (size_t q = 0; q < np && !awaited; q++) {
        const LaunchRec* hq = head(q);
        if (q == p || !hq || same(hq, h)) continue;
        for (size_t k = cur[q] + 1; k < R.parts[q].size() && !awaited; k++) awaited = same(&R.parts[q][k], h);
      }
      if (!awaited) lead = p;
    }
    if (lead == np)
      for (size_t p = 0; p < np && lead == np; p++)
        if (head(p)) lead = p;
    if (lead == np) return GS_OK;
    const LaunchRec* L = head(lead);
    const LaunchRec* grp[MAX_SEG];
    int ng = 0;
    for (size_t p = 0; p < np && ng < MAX_SEG; p++) {
      const LaunchRec* h = head(p);
      if (h && same(h, L)) {
        grp[ng++] = h;
        cur[p]++;
      }
    }
    RC(L->go(c, grp, ng));
  }
}

// ---------------------------------------------------------------------------
// shape helpers
// ---------------------------------------------------------------------------
// scalar preparation goes one-lane-per-output once an equation carries this many Gamma entries
static inline bool wide_prep(int m, int n) { return (long)m * n >= 1024; }
static inline bool x_is_group(int ty) { return ty == GS_PPE || ty == GS_MSMEG1; }
static inline bool y_is_group(int ty) { return ty == GS_PPE || ty == GS_MSMEG2; }

template <class C> struct Sz {
  // BOUNDARY sizes (include/gs_amd.h), not the internal radix-2^28 structs
  static constexpr size_t FQ = 4 * C::N, FR = sizeof(Fr<C>), G1 = 2 * FQ, G2 = 4 * FQ, GT = 12 * FQ,
                          COM1 = 2 * G1, COM2 = 2 * G2, CRS = 2 * COM1 + 2 * COM2 + G1 + G2 + GT;
};

// table ids inside tab16_g1/tab16_g2 (and the first-level tab_g1/tab_g2): 0 u0.0, 1 u0.1, 2 u1.0, 3 u1.1, 4 W.1 ; W.0 aliases u1.0
static inline uint8_t tb_u(int k, int c) { return (uint8_t)(2 * k + c); }
static inline uint8_t tb_w(int c) { return c ? 4 : 2; }

// Task tables are immutable once uploaded: they are cached per (name, size, content hash) so that a later batch never
// overwrites a table an in-flight kernel reads.  The host bytes are kept and compared on a hit (a hash collision gets
// its own entry instead of silently reusing the wrong plan); the cache is capped: past GS_PLAN_CAP entries the least
// recently used ones are freed after a stream sync (shapes seen long ago by a long-lived context).
constexpr size_t GS_PLAN_CAP = 256;
static int plan_evict(gs_ctx* c) {
  if (c->plans.size() <= GS_PLAN_CAP) return GS_OK;
  if (c->rec) return GS_OK;  // recorded launches hold pointers into the cached tables: evict on a later call
  hipStreamSynchronize(c->stream);
  for (hipStream_t st : c->side)
    if (st) hipStreamSynchronize(st);
  while (c->plans.size() > GS_PLAN_CAP / 2) {
    auto victim = c->plans.begin();
    for (auto it = c->plans.begin(); it != c->plans.end(); ++it)
      if (it->second.last_use < victim->second.last_use) victim = it;
    if (victim->second.dev.p) hipFree(victim->second.dev.p);
    c->plans.erase(victim);
  }
  return GS_OK;
}
template <class T> static int upload(gs_ctx* c, const char* name, const std::vector<T>& v, const T** out) {
  uint64_t h = 1469598103934665603ull;
  const uint8_t* b = (const uint8_t*)v.data();
  const size_t nb = v.size() * sizeof(T);
  {  // 8 bytes per step (large-arity task tables are megabytes; the content is compared on a hit anyway)
    size_t i = 0;
    for (; i + 8 <= nb; i += 8) {
      uint64_t w;
      memcpy(&w, b + i, 8);
      h = (h ^ w) * 1099511628211ull;
      h ^= h >> 29;
    }
    for (; i < nb; i++) h = (h ^ b[i]) * 1099511628211ull;
  }
  for (int salt = 0;; salt++) {
    char key[176];
    snprintf(key, sizeof key, "plan%s.%zu.%016llx.%d", name, v.size(), (unsigned long long)h, salt);
    auto it = c->plans.find(key);
    if (it != c->plans.end()) {
      PlanEntry& e = it->second;
      if (e.host.size() != nb || (nb && memcmp(e.host.data(), b, nb) != 0)) continue;  // collision: next salt
      if (!e.dev.p) {  // (cannot happen since entries are inserted after their upload; never hand out a null table)
        c->plans.erase(it);
        continue;
      }
      e.last_use = ++c->plan_clock;
      *out = (const T*)e.dev.p;
      return GS_OK;
    }
    RC(plan_evict(c));
    // the entry goes into the cache only once its device copy exists: a failed allocation or copy must not leave a
    // key behind that a later call (same shape, e.g. a retry with a smaller batch) would hit with a null table
    PlanEntry e;
    RC(ensure(c, e.dev, nb + 16));
    if (nb) {
      hipError_t ce = hipMemcpy(e.dev.p, b, nb, hipMemcpyHostToDevice);
      if (ce != hipSuccess) {
        hipFree(e.dev.p);
        return fail(c, GS_ERR_DEVICE, "task table upload", ce);
      }
    }
    e.host.assign(b, b + nb);
    e.last_use = ++c->plan_clock;
    *out = (const T*)e.dev.p;
    c->plans.emplace(key, std::move(e));
    return GS_OK;
  }
}

// ---------------------------------------------------------------------------
// one side (G1 or G2) of commit / prove expressed as engine tasks
// ---------------------------------------------------------------------------
struct SidePlan {
  std::vector<GrpTask> grp;  // when tm > 1: lanes of k_var_multi (each sums <= tm consecutive var tasks)
  int tm = 1;
  int mo = 1, w = 4;  // outputs per lane after share_tables(), window width
  bool shared_done = false;
  std::vector<VarTask> var;
  std::vector<FixTask> fix;
  std::vector<RedTask> red;
  int nslots = 0;
  int ncom = 0;  // the first `ncom` reductions (the commitments) only sum fixed-base partial slots
  int tab_bases = 0;  // > 0: every variable-base term is over array 0, which holds this many bases per equation, and
                      // the side runs on shared per-base window tables (k_tab_build + k_var_tab) instead of Straus lanes
};

static FixTask mkfix(int s0, int t0, int s1, int t1, int a_arr, int a_idx, int slot, int a_neg = 0) {
  FixTask f;
  f.s0 = (uint32_t)s0;
  f.s1 = (uint32_t)s1;
  f.t0 = (uint8_t)t0;
  f.t1 = (uint8_t)t1;
  f.a_arr = (uint8_t)a_arr;
  f.a_idx = (uint32_t)a_idx;
  f.a_neg = (uint8_t)a_neg;
  f.slot = (uint32_t)slot;
  return f;
}
static VarTask mkvar(int s, int arr, int idx, int slot) {
  VarTask v;
  v.s_idx = (uint32_t)s;
  v.p_arr = (uint8_t)arr;
  v.p_idx = (uint32_t)idx;
  v.slot = (uint32_t)slot;
  v.neg = 0;
  v.pad0 = v.pad1 = 0;
  return v;
}
static RedTask mkred(int b0, int e0, int b1, int e1, int out_arr, int out_idx) {
  RedTask r;
  r.b0 = (uint32_t)b0;
  r.e0 = (uint32_t)e0;
  r.b1 = (uint32_t)b1;
  r.e1 = (uint32_t)e1;
  r.out_arr = (uint8_t)out_arr;
  r.out_idx = (uint32_t)out_idx;
  r.pad = r.pad1 = r.pad2 = 0;
  return r;
}

// Append the variable-base terms of ONE output point: chunks of <= tm terms share a lane
// (and one partial slot); with tm == 1 every term is its own lane/slot.
static void add_var_terms(SidePlan& sp, const std::vector<VarTask>& terms, int& slot) {
  size_t T = terms.size();
  if (T == 0) return;
  size_t tm = sp.tm < 1 ? 1 : (size_t)sp.tm, ng = (T + tm - 1) / tm, base = T / ng, rem = T % ng, s0 = 0;
  for (size_t k = 0; k < ng; k++) {  // balanced groups: sizes differ by <= 1
    size_t n = base + (k < rem ? 1 : 0);
    GrpTask g;
    memset(&g, 0, sizeof g);
    g.first[0] = (uint32_t)sp.var.size();
    g.nt = (uint32_t)n;
    g.no = 1;
    g.slot[0] = (uint32_t)slot;
    for (size_t i = 0; i < n; i++) {
      VarTask v = terms[s0 + i];
      v.slot = (uint32_t)slot;
      sp.var.push_back(v);
    }
    s0 += n;
    sp.grp.push_back(g);
    slot++;
  }
}
// Lane cost of the joint MSM in Fq multiplications: one table build of `nt` bases (2^(w-1) entries each: doublings,
// mixed additions, 7 multiplications per entry for the common denominator) + `no` main loops (w doublings per window,
// one mixed addition per non-zero digit, the endomorphism on the looked-up entry).
static double straus_lane_cost(bool g2, bool bn, int nt, int no, int w) {
  const double madd = g2 ? 29.0 : 11.0, dbl = g2 ? 16.0 : 8.0, gz = g2 ? 21.0 : 7.0, endo = g2 ? 4.0 : 0.5;
  const int ns = g2 ? 4 : 2, nl = g2 ? (bn ? 3 : 2) : (bn ? 5 : 4), nd = (32 * nl + w - 1) / w + 1, ne = 1 << (w - 1);
  double build = nt * (ne / 2 * dbl + (ne / 2 - 1) * madd + ne * gz);
  double run = (nd - 1) * w * dbl + (double)nt * ns * (nd - 0.5) * (1.0 - 1.0 / (2 * ne)) * (madd + endo);
  return build + no * run;
}
// Variable-base terms per lane (joint Straus MSM, shared doubling chain) for a side with `T` terms per output
// and `outputs` outputs per equation.  Same cost model as miller_cost: rounds of waves x lane length, lane(nt) =
// D + P nt Fq multiplications (fits of profiles/r1/fq_mul_counts.json, the same within 3 % on both curves now that
// both have endomorphism decompositions: G1 832 + 829 nt, G2 1000 + 2365 nt).
// Lane-time multiplier of a launch of `waves` waves.  The Fp2 / Fp12 kernels and the BLS12-381 G1 kernels hold one
// wave per SIMD: whole rounds while there are few.  The BN254 G1 Straus kernels come out of the compiler at 252
// registers and run two waves per SIMD; two resident waves finish in 2 / 1.55 of the time of one (measured at 2^16:
// k_var_multi8.g1 11.4 ms as 2048 waves against 16.7 ms as 1024 waves of twice the length).
static double wave_rounds(const gs_ctx* c, double waves, bool g2) {
  double r = waves / (double)c->simd_slots;
  if (g2 || c->curve != 1) {
    if (r <= 1.0) return 1.0;
    return r < 4.0 ? (double)(size_t)(r + 0.999) : r;
  }
  const double gain = 1.55;
  return r <= gain ? 1.0 : r / gain;
}
static int pick_tm(const gs_ctx* c, size_t N, int T, int outputs, bool g2, int share) {
  // `share` of the `outputs` run over the same bases (share_tables() can give them one table build per lane): the
  // group size is chosen together with the outputs per lane and the window width that share_tables() will then pick
  if (T < 2 || !c->endo) return 1;  // (no endomorphisms: one plain lane per term)
  if (c->var_tm > 0) {  // forced: the balanced group size is what runs
    int ng = (T + c->var_tm - 1) / c->var_tm;
    return (T + ng - 1) / ng;
  }
  int best_tm = 1;
  double best = -1;
  for (int tm = 1; tm <= 8; tm++) {
    int ng = (T + tm - 1) / tm, eff = (T + ng - 1) / ng;
    if (eff != tm) continue;  // the balanced size is what runs; skip aliases
    for (int mo : {1, 2, 4}) {
      if (mo > 1 && (eff < 2 || mo > share)) continue;  // one-term lanes are k_var: no sharing
      if (c->var_mo > 0 && mo != c->var_mo && !(eff < 2 && mo == 1)) continue;
      for (int w : {4, 5}) {
        if (w == 5 && eff < 2) continue;
        if (c->var_w > 0 && w != c->var_w && eff >= 2) continue;
        double lanes = (double)N * ((double)outputs / mo) * ng;
        double rounds = wave_rounds(c, lanes / 64.0, g2);
        // + the lane of k_red that folds the ng partial sums of an output (matters for large arities)
        double cost = rounds * straus_lane_cost(g2, c->curve == 1, eff, mo, w) +
                      fold_cost((double)ng, g2 ? 29.0 + 14.0 : 16.0);
        if (best < 0 || cost < best) {
          best = cost;
          best_tm = eff;
        }
      }
    }
  }
  return best_tm;
}

// Groups with the same bases (same arrays, indices and signs in the same order) become ONE lane of up to `mo` outputs
// that builds its tables once: the proof elements of a side share their constants and variables, the columns of
// Gamma^T c share the commitments.  Picks (mo, w) by the cost model of pick_tm unless overridden.
static void share_tables(const gs_ctx* c, size_t N, SidePlan& sp, bool g2) {
  if (sp.shared_done) return;
  sp.shared_done = true;
  sp.mo = 1;
  sp.w = 4;
  if (sp.tab_bases > 0) return;  // the lanes read the bases' shared tables: nothing to build per lane, one output each
  if (sp.tm <= 1 || sp.grp.empty()) return;
  // families of groups over the same bases, in plan order: hash of (nt, (index, array, sign) of every base), with
  // the bases compared on a hash match (large arities have ~10^5 groups: no strings, no ordered map)
  auto same = [&](const GrpTask& a, const GrpTask& b) {
    if (a.nt != b.nt) return false;
    for (uint32_t i = 0; i < a.nt; i++) {
      const VarTask &x = sp.var[a.first[0] + i], &y = sp.var[b.first[0] + i];
      if (x.p_idx != y.p_idx || x.p_arr != y.p_arr || x.neg != y.neg) return false;
    }
    return true;
  };
  std::vector<std::vector<size_t>> fam;
  std::unordered_map<uint64_t, std::vector<size_t>> by_hash;  // hash -> families
  int ntmax = 1;
  for (size_t i = 0; i < sp.grp.size(); i++) {
    const GrpTask& g = sp.grp[i];
    uint64_t h = 1469598103934665603ull ^ g.nt;
    for (uint32_t k = 0; k < g.nt; k++) {
      const VarTask& v = sp.var[g.first[0] + k];
      h = (h ^ (((uint64_t)v.p_idx << 16) | ((uint64_t)v.p_arr << 8) | v.neg)) * 1099511628211ull;
    }
    std::vector<size_t>& cand = by_hash[h];
    size_t f = (size_t)-1;
    for (size_t c2 : cand)
      if (same(sp.grp[fam[c2][0]], g)) {
        f = c2;
        break;
      }
    if (f == (size_t)-1) {
      f = fam.size();
      fam.emplace_back();
      cand.push_back(f);
    }
    fam[f].push_back(i);
    ntmax = std::max(ntmax, (int)g.nt);
  }
  size_t share = 1;
  for (auto& f : fam) share = std::max(share, f.size());
  int best_mo = 1, best_w = 4;
  double best = -1;
  for (int mo : {1, 2, 4}) {
    if (c->var_mo > 0 && mo != c->var_mo) continue;
    if (c->var_mo <= 0 && (size_t)mo > share) continue;
    size_t lanes = 0;
    for (auto& f : fam) lanes += (f.size() + mo - 1) / mo;
    for (int w : {4, 5}) {
      if (c->var_w > 0 && w != c->var_w) continue;
      // a part of a mixed call: one window width for every part, or their Straus lanes are two kernel instances and go
      // out as two under-filled launches (2^12 mixed: 3.1 + 4.7 ms where one launch takes 4.7)
      if (c->var_w <= 0 && c->rec && w != 4) continue;
      double rounds = wave_rounds(c, (double)N * lanes / 64.0, g2);
      int eff_mo = (int)std::min((size_t)mo, share);
      double cost = rounds * straus_lane_cost(g2, c->curve == 1, ntmax, eff_mo, w);
      if (best < 0 || cost < best) {
        best = cost;
        best_mo = mo;
        best_w = w;
      }
    }
  }
  sp.mo = best_mo;
  sp.w = best_w;
  if (getenv("GS_PLAN_TRACE")) {
    size_t lanes = 0;
    for (auto& f : fam) lanes += (f.size() + best_mo - 1) / best_mo;
    fprintf(stderr, "[plan] straus %s fill %zu: %zu groups in %zu families (largest %zu), <= %d terms; outputs per build %d, window %d, "
            "%zu lanes per equation\n", g2 ? "G2" : "G1", N, sp.grp.size(), fam.size(), share, ntmax, best_mo, best_w, lanes);
  }
  if (best_mo == 1) return;
  std::vector<GrpTask> merged;
  for (const std::vector<size_t>& f : fam) {
    for (size_t b = 0; b < f.size(); b += best_mo) {
      GrpTask g = sp.grp[f[b]];
      g.no = (uint32_t)std::min((size_t)best_mo, f.size() - b);
      for (uint32_t o = 1; o < g.no; o++) {
        g.first[o] = sp.grp[f[b + o]].first[0];
        g.slot[o] = sp.grp[f[b + o]].slot[0];
      }
      merged.push_back(g);
    }
  }
  sp.grp.swap(merged);
}

// Build the plan of one side.
//  nv      committed variables on this side (m for the G1 side, n for the G2 side)
//  nc      constants paired with the OTHER side's variables that live here (len of A for G1 side = n; B for G2 = m)
//  group   variables on this side are group elements (else scalars)
//  kc      columns of this side's commit randomness (2 group / 1 scalar)
//  npf     number of proof elements on this side (theta count = ky for G1 side, pi count = kx for G2 side)
//  offsets into the pool: rc (nv x kc commit randomness), vc (scalar variables), cs (nc x npf scalars multiplying
//  the constants: SC for the G1 side / RC for the G2 side, indexed [j*npf + l]), ph (npf x nv: PHI / PSI),
//  f0 (npf x kc fixed scalars: TC for G1 side, OM for G2 side), sg (npf: SIG / RHO)
//  arrays: 0 = variables (group), 1 = constants (group)
static void build_side(SidePlan& sp, bool want_coms, int nv, int nc, bool group, int kc, int npf, int rc, int vc,
                       int cs, int ph, int f0, int sg) {
  int slot = 0;
  if (want_coms) {
    for (int i = 0; i < nv; i++) {
      int s_begin = slot;
      for (int c = 0; c < 2; c++) {
        if (group)
          sp.fix.push_back(mkfix(rc + 2 * i, tb_u(0, c), rc + 2 * i + 1, tb_u(1, c), c ? 0 : 0xFF, i, slot++));
        else
          sp.fix.push_back(mkfix(vc + i, tb_w(c), rc + i, tb_u(0, c), 0xFF, 0, slot++));
      }
      sp.red.push_back(mkred(s_begin, s_begin + 1, s_begin + 1, s_begin + 2, 0, i));
    }
    sp.ncom = nv;
  }
  for (int l = 0; l < npf; l++) {
    int b0 = slot;
    if (group) {
      sp.fix.push_back(mkfix(f0 + l * 2, tb_u(0, 0), f0 + l * 2 + 1, tb_u(1, 0), 0xFF, 0, slot++));
      int b1 = slot;
      sp.fix.push_back(mkfix(f0 + l * 2, tb_u(0, 1), f0 + l * 2 + 1, tb_u(1, 1), 0xFF, 0, slot++));
      std::vector<VarTask> terms;
      for (int j = 0; j < nc; j++) terms.push_back(mkvar(cs + j * npf + l, 1, j, 0));
      for (int i = 0; i < nv; i++) terms.push_back(mkvar(ph + l * nv + i, 0, i, 0));
      add_var_terms(sp, terms, slot);
      sp.red.push_back(mkred(b0, b1, b1, slot, 1, l));
    } else {
      sp.fix.push_back(mkfix(sg + l, tb_w(0), f0 + l, tb_u(0, 0), 0xFF, 0, slot++));
      int b1 = slot;
      sp.fix.push_back(mkfix(sg + l, tb_w(1), f0 + l, tb_u(0, 1), 0xFF, 0, slot++));
      sp.red.push_back(mkred(b0, b1, b1, slot, 1, l));
    }
  }
  sp.nslots = slot;
}

// the reduction launch of a side, kept back so that the caller can run the two sides' reductions side by side
struct RedLaunch {
  std::string name, tag;
  std::vector<RedTask> hred;  // host copy: long runs are folded in parallel first (run_red)
  const void* part = nullptr;
  size_t nred = 0;
  int nslots = 0;
  OutTab outs;
};
constexpr uint32_t RED_FOLD_K = 8, RED_RUN_MAX = 16;
template <class C, class F> static int run_red(gs_ctx* c, size_t N, const RedLaunch& r) {
  std::vector<RedTask> red = r.hred;
  const Jac<F>* part = (const Jac<F>*)r.part;
  int nslots = r.nslots;
  // segmented K-ary tree: while a component of some output still sums more than RED_RUN_MAX slots, fold runs of K
  for (int level = 0;; level++) {
    uint32_t longest = 0;
    for (const RedTask& t : red) {
      longest = std::max(longest, t.e0 - t.b0);
      longest = std::max(longest, t.e1 - t.b1);
    }
    if (longest <= RED_RUN_MAX) break;
    std::vector<FoldTask> ft;
    uint32_t next = 0;
    auto fold_range = [&](uint32_t& b, uint32_t& e) {
      uint32_t nb = next;
      for (uint32_t lo = b; lo < e; lo += RED_FOLD_K) ft.push_back(FoldTask{lo, std::min(lo + RED_FOLD_K, e), next++, 0});
      b = nb;
      e = next;
    };
    for (RedTask& t : red) {
      fold_range(t.b0, t.e0);
      fold_range(t.b1, t.e1);
    }
    const FoldTask* dft;
    RC(upload(c, (r.tag + ".fold").c_str(), ft, &dft));
    void* out;
    RC(scratch(c, (r.tag + (level & 1 ? ".fold1" : ".fold0")).c_str(), N * (size_t)next * sizeof(Jac<F>), &out));
    c->work_hint = N * (uint64_t)nslots;
    RC(launch_seg<k_slot_fold<C, F>>(c, (std::string("k_slot_fold") + r.tag).c_str(), N * ft.size(), 64, N * ft.size(),
              (int)ft.size(), dft, part, nslots, (Jac<F>*)out, (int)next));
    part = (const Jac<F>*)out;
    nslots = (int)next;
  }
  const RedTask* dred;
  RC(upload(c, (r.tag + ".red").c_str(), red, &dred));
  c->work_hint = N * (uint64_t)nslots;  // partial sums folded
  // outputs per lane (one inversion each lane): as many as still leave the launch a full round of waves
  int K = 1;
  // (K may exceed the outputs of an equation: 6 outputs go to ONE lane of 8 rather than to lanes of 4 + 2 with an
  // inversion each, once a lane per equation still fills the chip)
  while (K < RED_K && K < (int)r.nred && (fillN(c, N) * ((r.nred + 2 * K - 1) / (2 * K)) + 63) / 64 >= c->simd_slots) K *= 2;
  if (c->red_k > 0) K = std::min(c->red_k, RED_K);
  size_t lanes = N * ((r.nred + K - 1) / K);
  return launch_seg<k_red<C, F>>(c, r.name.c_str(), lanes, 64, lanes, (int)r.nred, dred, part, nslots, r.outs, K);
}
template <class C, class F>
static int run_side(gs_ctx* c, const char* tag, size_t N, SidePlan& sp, const ArrTab& arrs, const Fr<C>* pool,
                    int pool_n, const Aff<F>* tab, const OutTab& outs, hipStream_t vstream = nullptr,
                    hipEvent_t vev0 = nullptr, hipEvent_t vev1 = nullptr, RedLaunch* defer = nullptr,
                    const unsigned* pipe_masks = nullptr) {
  // pipe_masks (host-pointer calls): {inputs the fixed-base kernel reads, inputs the variable-base kernel reads,
  // outputs complete after the commitments' reduction, outputs complete after the side's last reduction}
  std::string t(tag);
  share_tables(c, fillN(c, N), sp, std::is_same<F, Fp2<C>>::value);
  const VarTask* dvar;
  const FixTask* dfix;
  RC(upload(c, (t + ".var").c_str(), sp.var, &dvar));
  RC(upload(c, (t + ".fix").c_str(), sp.fix, &dfix));
  void* part;
  RC(scratch(c, (t + ".part").c_str(), N * sp.nslots * sizeof(Jac<F>), &part));
  // fork: the variable-base kernel goes to `vstream` (if any) while the fixed-base one stays on the side's stream
  hipStream_t base = c->cur ? c->cur : c->stream;
  bool fork = vstream != nullptr && !c->prof;
  if (fork) {
    hipEventRecord(vev0, base);
    hipStreamWaitEvent(vstream, vev0, 0);
  }
  if (pipe_masks) RC(need(c, pipe_masks[0]));
  {
    uint64_t terms = 0;  // fixed-base scalars per equation
    for (const FixTask& f : sp.fix) terms += (f.t0 != 0xFF) + (f.t1 != 0xFF);
    c->work_hint = N * terms;
  }
  RC(launch_seg<k_fix<C, F>>(c, (std::string("k_fix") + tag).c_str(), N * sp.fix.size(), 64, N * sp.fix.size(),
            (int)sp.fix.size(), dfix, arrs, pool, pool_n, tab, (Jac<F>*)part, sp.nslots));
  // host-pointer calls: the commitments depend on the fixed-base kernel alone -- reduce them now and send them back
  // across PCIe under the variable-base kernel
  // (from 2^14 equations on: below that the extra launch and events cost more than the copy they hide)
  const bool early = c->pipe && pipe_masks && !c->rec && !fork && !defer && sp.ncom > 0 && sp.ncom < (int)sp.red.size() &&
                     N >= 16 * c->simd_slots;
  if (early) {
    RedLaunch rc0;
    rc0.name = std::string("k_red") + tag;
    rc0.tag = t + ".coms";
    rc0.hred.assign(sp.red.begin(), sp.red.begin() + sp.ncom);
    rc0.part = part;
    rc0.nred = (size_t)sp.ncom;
    rc0.nslots = sp.nslots;
    rc0.outs = outs;
    RC((run_red<C, F>(c, N, rc0)));
    RC(out_ready(c, pipe_masks[2]));
  }
  if (fork) c->cur = vstream;
  // (the variable-base kernel reads the variables too, and on a forked stream nothing has waited for them yet)
  if (pipe_masks) RC(need(c, pipe_masks[0] | pipe_masks[1]));
  if (sp.tab_bases > 0) {
    // shared per-base window tables (8-bit signed windows: 128 affine multiples per base), then Straus-shaped lanes of
    // up to 8 terms that only read them
    constexpr int TW = 8, TNE = 1 << (TW - 1);
    const size_t nbt = N * (size_t)sp.tab_bases;
    void *tabs, *stage;
    RC(scratch(c, (t + ".tab8").c_str(), nbt * TNE * sizeof(Aff<F>), &tabs));
    RC(scratch(c, (t + ".tab8j").c_str(), nbt * TNE * sizeof(Jac<F>), &stage));
    c->work_hint = nbt;
    RC((launch_seg<k_tab_build<C, F, TW>>(c, (std::string("k_tab_build") + tag).c_str(), nbt, 64, nbt, sp.tab_bases, arrs, 0,
                                          (Jac<F>*)stage, (Aff<F>*)tabs)));
    const GrpTask* dgrp;
    RC(upload(c, (t + ".grp").c_str(), sp.grp, &dgrp));
    const size_t tot = N * sp.grp.size();
    c->work_hint = N * sp.var.size();  // terms
    RC((launch_seg<k_var_tab<C, F, 8, TW>>(c, (std::string("k_var_tab8") + tag).c_str(), tot, 64, tot, (int)sp.grp.size(), dgrp,
                                           dvar, pool, pool_n, (Jac<F>*)part, sp.nslots, (const Aff<F>*)tabs, sp.tab_bases)));
  } else if (sp.tm <= 1 && !c->endo) {
    RC((launch_seg<k_var<C, F, false>>(c, (std::string("k_var.plain") + tag).c_str(), N * sp.var.size(), 64, N * sp.var.size(),
              (int)sp.var.size(), dvar, arrs, pool, pool_n, (Jac<F>*)part, sp.nslots)));
  } else if (sp.tm <= 1) {
    RC(launch_seg<k_var<C, F>>(c, (std::string("k_var") + tag).c_str(), N * sp.var.size(), 64, N * sp.var.size(),
              (int)sp.var.size(), dvar, arrs, pool, pool_n, (Jac<F>*)part, sp.nslots));
  } else {
    const GrpTask* dgrp;
    RC(upload(c, (t + ".grp").c_str(), sp.grp, &dgrp));
    size_t tot = N * sp.grp.size();
    c->work_hint = N * sp.var.size();  // terms
    // the lanes' Straus tables: lane-contiguous global workspace (see jac_msm_straus_at), at most VAR_WS_LANES lanes
    // of it; a larger batch goes in several launches over the same workspace
    // (parts of a mixed call share a launch when they agree on the instance: the window width is fixed for them in
    // share_tables; the 4- or 8-term instance follows the part's own group size -- the 8-term one runs 4-term groups
    // too, but 1.8x slower (2^12 mixed: 5.6 against 3.1 ms), so parts are not forced onto it)
    const int tmax = sp.tm > 4 ? 8 : 4;
    const size_t lane_ws = ((size_t)tmax << (sp.w - 1)) * (sizeof(Aff<F>) + sizeof(Jac<F>));
    // (the buffer is grow-only: a workspace that already holds more lanes is used as it is)
    size_t have = 0;
    {
      auto it = c->scratch.find((c->scratch_tag ? "m" + std::to_string(c->scratch_tag) + ":" : std::string()) + t + ".tabws");
      if (it != c->scratch.end()) have = it->second.cap / lane_ws;
    }
    const size_t chunk = std::min(tot, c->var_ws_lanes > 0 ? (size_t)c->var_ws_lanes
                                                            : std::max(std::min(have, tot), var_ws_auto(c, tot, lane_ws)));
    void* tabws;
    RC(scratch(c, (t + ".tabws").c_str(), chunk * lane_ws, &tabws));
    // kernel name: k_var_multi<TMAX>[w5][x<outputs per lane>]
    std::string kn = std::string("k_var_multi") + (tmax == 4 ? "4" : "8") + (sp.w == 5 ? "w5" : "") +
                     (sp.mo > 1 ? "x" + std::to_string(sp.mo) : "") + tag;
    for (size_t g0 = 0; g0 < tot; g0 += chunk) {
      size_t n = std::min(chunk, tot - g0);
      c->work_hint = (uint64_t)((double)N * sp.var.size() * ((double)n / (double)tot));  // terms
      // G1 lanes in the two-waves-per-SIMD build when the launch puts (nearly) two waves on every SIMD: fewer, and the
      // dispatcher doubles waves up on some SIMDs while others idle (measured in round 2: 27 ms instead of 17.7)
      // MEASURED (round 4, gpurun_out/r4h): the verifier's 2048-wave launch at 2^16 takes 28.85 ms in the two-wave
      // build against 29.07 ms in the one-wave build -- the multiply-add pipe is quarter rate whoever issues into
      // it, and the 256-register build pays for the waits it hides with spills.  Kept as a forced shape (var_w2 = 1),
      // never planned.
      const bool w2 = std::is_same<F, Fq<C>>::value && c->var_w2 == 1;
#define GS_VM(TM, WW)                                                                                                \
  do {                                                                                                               \
    if (w2)                                                                                                          \
      RC((launch_seg<k_var_multi<C, F, TM, WW, (std::is_same<F, Fq<C>>::value ? 2 : 1)>>(                            \
          c, kn.c_str(), n, 64, tot, (int)sp.grp.size(), dgrp, dvar, arrs, pool, pool_n, (Jac<F>*)part, sp.nslots, g0, \
          (Aff<F>*)tabws)));                                                                                         \
    else                                                                                                             \
      RC((launch_seg<k_var_multi<C, F, TM, WW>>(c, kn.c_str(), n, 64, tot, (int)sp.grp.size(), dgrp, dvar, arrs, pool, \
                                                pool_n, (Jac<F>*)part, sp.nslots, g0, (Aff<F>*)tabws)));              \
  } while (0)
      if (tmax == 4 && sp.w == 5)
        GS_VM(4, 5);
      else if (tmax == 4)
        GS_VM(4, 4);
      else if (sp.w == 5)
        GS_VM(8, 5);
      else
        GS_VM(8, 4);
#undef GS_VM
    }
  }
  if (fork) {  // join before the reduction
    hipEventRecord(vev1, vstream);
    c->cur = base == c->stream ? nullptr : base;
    hipStreamWaitEvent(base, vev1, 0);
  }
  RedLaunch r;
  r.name = std::string("k_red") + tag;
  r.tag = t;
  if (early)
    r.hred.assign(sp.red.begin() + sp.ncom, sp.red.end());
  else
    r.hred = sp.red;
  r.part = part;
  r.nred = r.hred.size();
  r.nslots = sp.nslots;
  r.outs = outs;
  if (defer) {
    *defer = r;
    return GS_OK;
  }
  RC((run_red<C, F>(c, N, r)));
  if (pipe_masks) RC(out_ready(c, pipe_masks[2] | pipe_masks[3]));
  return GS_OK;
}

// ---------------------------------------------------------------------------
// per-curve implementation
// ---------------------------------------------------------------------------
template <class C> struct Impl {
  typedef Fq<C> F1;
  typedef Fp2<C> F2;
  typedef Aff<F1> A1;
  typedef Aff<F2> A2;
  typedef Fr<C> S;
  typedef Fp12<C> GT;
  typedef Sz<C> Z;

  static int set_crs(gs_ctx* c, const void* crs_host) {
    const uint8_t* h = (const uint8_t*)crs_host;
    // an identical CRS already installed on this device (by any context of the process)?  share its tables
    {
      std::lock_guard<std::mutex> lk(g_tables_mu);
      for (auto it = g_tables.begin(); it != g_tables.end();) {
        std::shared_ptr<CrsTables> t = it->lock();
        if (!t) {
          it = g_tables.erase(it);
          continue;
        }
        if (t->device == c->device && t->curve == c->curve && t->crs_bytes.size() == Z::CRS &&
            memcmp(t->crs_bytes.data(), h, Z::CRS) == 0) {
          c->tabs = t;
          c->have_crs = true;
          return GS_OK;
        }
        ++it;
      }
    }
    std::shared_ptr<CrsTables> tb = std::make_shared<CrsTables>();
    tb->device = c->device;
    tb->curve = c->curve;
    tb->crs_bytes.assign(h, h + Z::CRS);
    c->have_crs = false;
    c->tabs = tb;  // (a failed build leaves have_crs false; the tables die with the last reference)
    // device copies (boundary form): 6 G1 points u0.0 u0.1 u1.0 u1.1 W1.0 W1.1, same for G2
    std::vector<uint8_t> g1pts(6 * Z::G1), g2pts(6 * Z::G2);
    memcpy(&g1pts[0], h, 4 * Z::G1);
    memcpy(&g2pts[0], h + 2 * Z::COM1, 4 * Z::G2);
    const uint8_t* g1 = h + 2 * Z::COM1 + 2 * Z::COM2;
    const uint8_t* g2 = g1 + Z::G1;
    // W1 = u[1] + (O, g1), W2 = v[1] + (O, g2)   (data_structures.rs:323-326, 368-371), derived on the device:
    // slot 4 = u1.0, slot 5 holds the addend on the way in
    memcpy(&g1pts[4 * Z::G1], &g1pts[2 * Z::G1], Z::G1);
    memcpy(&g2pts[4 * Z::G2], &g2pts[2 * Z::G2], Z::G2);
    memcpy(&g1pts[5 * Z::G1], g1, Z::G1);
    memcpy(&g2pts[5 * Z::G2], g2, Z::G2);
    RC(ensure(c, tb->crs_g1, g1pts.size()));
    RC(ensure(c, tb->crs_g2, g2pts.size()));
    HIPCHK(c, hipMemcpy(tb->crs_g1.p, g1pts.data(), g1pts.size(), hipMemcpyHostToDevice));
    HIPCHK(c, hipMemcpy(tb->crs_g2.p, g2pts.data(), g2pts.size(), hipMemcpyHostToDevice));
    RC(launch(c, "k_crs_derive.g1", k_crs_derive<C, F1>, 1, 64, (uint8_t*)tb->crs_g1.p));
    RC(launch(c, "k_crs_derive.g2", k_crs_derive<C, F2>, 1, 64, (uint8_t*)tb->crs_g2.p));
    // Miller line tables of the six G2 points (v0, v1, W2): the verifier pairs theta / PB against them
    RC(ensure(c, tb->line_tab, 6 * (size_t)miller_line_count<C>() * sizeof(Line<C>)));
    RC(launch(c, "k_line_tables", k_line_tables<C>, 6, 64, (const uint8_t*)tb->crs_g2.p, (Line<C>*)tb->line_tab.p));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipMemcpy(g1pts.data(), tb->crs_g1.p, g1pts.size(), hipMemcpyDeviceToHost));
    HIPCHK(c, hipMemcpy(g2pts.data(), tb->crs_g2.p, g2pts.size(), hipMemcpyDeviceToHost));
    // window tables (internal form) for bases {u0.0,u0.1,u1.0,u1.1,W.1}
    std::vector<uint8_t> b1(5 * Z::G1), b2(5 * Z::G2);
    memcpy(&b1[0], &g1pts[0], 4 * Z::G1);
    memcpy(&b1[4 * Z::G1], &g1pts[5 * Z::G1], Z::G1);
    memcpy(&b2[0], &g2pts[0], 4 * Z::G2);
    memcpy(&b2[4 * Z::G2], &g2pts[5 * Z::G2], Z::G2);
    void *db1, *db2;
    RC(scratch(c, "crs.b1", b1.size(), &db1));
    RC(scratch(c, "crs.b2", b2.size(), &db2));
    HIPCHK(c, hipMemcpy(db1, b1.data(), b1.size(), hipMemcpyHostToDevice));
    HIPCHK(c, hipMemcpy(db2, b2.data(), b2.size(), hipMemcpyHostToDevice));
    // first-level (8-bit) window tables only feed k_build_tables16: context scratch, reused by the next CRS
    size_t ne = (size_t)5 * 32 * 256;
    void *t8g1, *t8g2;
    RC(scratch(c, "crs.tab8.g1", ne * sizeof(A1), &t8g1));
    RC(scratch(c, "crs.tab8.g2", ne * sizeof(A2), &t8g2));
    RC(launch(c, "k_build_tables.g1", k_build_tables<C, F1>, ne, 64, 5, (const uint8_t*)db1, (A1*)t8g1));
    RC(launch(c, "k_build_tables.g2", k_build_tables<C, F2>, ne, 64, 5, (const uint8_t*)db2, (A2*)t8g2));
    size_t ne16 = (size_t)5 * 16 * 65536;
    RC(ensure(c, tb->tab16_g1, ne16 * sizeof(A1)));
    RC(ensure(c, tb->tab16_g2, ne16 * sizeof(A2)));
    RC(launch(c, "k_build_tables16.g1", k_build_tables16<C, F1>, ne16 / 16, 64, 5, (const A1*)t8g1, (A1*)tb->tab16_g1.p));
    RC(launch(c, "k_build_tables16.g2", k_build_tables16<C, F2>, ne16 / 16, 64, 5, (const A2*)t8g2, (A2*)tb->tab16_g2.p));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    {
      std::lock_guard<std::mutex> lk(g_tables_mu);
      g_tables.push_back(tb);
    }
    c->have_crs = true;
    return GS_OK;
  }

  static PoolMap prove_pool(int m, int n, int kx, int ky) {
    PoolMap pm;
    memset(&pm, 0, sizeof pm);
    int o = 0;
    pm.RC = o; o += m * kx;
    pm.SC = o; o += n * ky;
    pm.PSI = o; o += kx * n;
    pm.PHI = o; o += ky * m;
    pm.OM = o; o += kx * ky;
    pm.TC = o; o += ky * kx;
    pm.RHO = o; o += kx;
    pm.SIG = o; o += ky;
    pm.XC = o; o += m;
    pm.YC = o; o += n;
    pm.total = o;
    return pm;
  }

  // prove / commit_and_prove for all four types (prove.rs:71-489)
  // shared: a Statement -- the N equations are over the SAME variables X, Y and commit randomness R, S (one copy,
  // stride 0); constants, Gamma, T and the proofs stay per equation.  Commitments are then made once by the caller.
  static int prove(gs_ctx* c, int ty, size_t N, int m, int n, const void* X, const void* Y, const void* A,
                   const void* B, const void* G, const void* R, const void* Sm, const void* T, void* xcoms,
                   void* ycoms, void* pi, void* theta, bool shared = false) {
    bool xg = x_is_group(ty), yg = y_is_group(ty);
    int kx = xg ? 2 : 1, ky = yg ? 2 : 1;
    PoolMap pm = prove_pool(m, n, kx, ky);
    void* pool;
    RC(scratch(c, "prove.pool", N * pm.total * sizeof(S), &pool));
    RangeGuard rg_all("gs.prove");
    // (host-pointer calls: the scalars first; group-valued variables and constants are not read before their side)
    RC(need(c, BIT(PI_G) | BIT(PI_R) | BIT(PI_S) | BIT(PI_T) | (xg ? 0u : BIT(PI_X) | BIT(PI_A)) |
                   (yg ? 0u : BIT(PI_Y) | BIT(PI_B))));
    if (wide_prep(m, n)) {  // large arity: one lane per output scalar
      int W = m * kx + n * ky + ky * kx + m + n + kx * n + ky * m;
      RC(launch_seg<k_prep_prove_wide_a<C>>(c, "k_prep_prove.a", N * (size_t)W, 64, N * (size_t)W, W, m, n, kx, ky,
                (const S*)G, (const S*)R, (const S*)Sm, (const S*)T, xg ? nullptr : (const S*)X,
                yg ? nullptr : (const S*)Y, pm, (S*)pool, shared ? 1 : 0));
      size_t tb = N * (size_t)(kx * ky + kx + ky);
      RC(launch_seg<k_prep_prove_wide_b<C>>(c, "k_prep_prove.b", tb, 64, tb, m, n, kx, ky, (const S*)R, (const S*)Sm,
                (const S*)T, xg ? nullptr : (const S*)X, yg ? nullptr : (const S*)Y, xg ? nullptr : (const S*)A,
                yg ? nullptr : (const S*)B, pm, (S*)pool, shared ? 1 : 0));
    } else {
      RC(launch_seg<k_prep_prove<C>>(c, "k_prep_prove", N, 64, N, m, n, kx, ky, (const S*)G, (const S*)R, (const S*)Sm,
                (const S*)T, xg ? nullptr : (const S*)X, yg ? nullptr : (const S*)Y, xg ? nullptr : (const S*)A,
                yg ? nullptr : (const S*)B, pm, (S*)pool, shared ? 1 : 0));
    }
    // small batches: G1 side on the context's stream, G2 side on side[0], each side's variable-base kernel on a
    // further stream; everything joins back before this function returns
    struct CurGuard {
      gs_ctx* c;
      ~CurGuard() { c->cur = nullptr; }
    } guard{c};
    // (measured: +18 % at 2^10, but -4 % at 2^12 where each variable-base kernel already fills the SIMDs, so only
    // while a side's variable-base lanes occupy at most half of them)
    const bool ov = c->overlap && !c->prof && !c->rec && c->side[0] && fillN(c, N) * (size_t)(m + n) * 2 <= 32 * c->simd_slots;
    if (ov) {
      hipEventRecord(c->sev[0], c->stream);
      hipStreamWaitEvent(c->side[0], c->sev[0], 0);
    }
    // larger batches: only the two reduction kernels (one inversion chain per lane, a few hundred waves each) run
    // side by side, after both sides' partial sums are complete
    // (not for a host-pointer call: there the G1 side's outputs go back across PCIe under the G2 side's kernels)
    // (round 3: OFF.  A side stream is another hardware queue with its own private-segment reservation; two of them
    // holding a few hundred MB each is what starved the main queue's big kernels of scratch waves later in the
    // process -- profiles/r3/scratch_pool.txt.  The 1-2 % it gained at 2^12..2^16 are not worth a 4x cliff.)
    const bool pair_reds = false;
    RedLaunch red1, red2;
    // G1 side: xcoms (m) + theta (ky).  constants A (len n) multiply S; Phi multiplies X; fixed part T.
    auto side_g1 = [&](bool first) -> int {
      SidePlan sp;
      sp.tm = xg ? pick_tm(c, fillN(c, N), m + n, ky, false, ky) : 1;
      build_side(sp, xcoms != nullptr, m, n, xg, kx, ky, pm.RC, pm.XC, pm.SC, pm.PHI, pm.TC, pm.SIG);
      ArrTab arrs;
      memset(&arrs, 0, sizeof arrs);
      if (xg) {
        arrs.base[0] = (const uint8_t*)X;
        arrs.stride[0] = shared ? 0u : (uint32_t)(m * Z::G1);
        arrs.base[1] = (const uint8_t*)A;
        arrs.stride[1] = (uint32_t)(n * Z::G1);
      }
      OutTab outs;
      memset(&outs, 0, sizeof outs);
      outs.base[0] = (uint8_t*)xcoms;
      outs.stride[0] = (uint32_t)(m * Z::COM1);
      outs.base[1] = (uint8_t*)theta;
      outs.stride[1] = (uint32_t)(ky * Z::COM1);
      RangeGuard rg("gs.prove.g1");
      // (the fixed-base kernel reads X as the commitments' affine addend; the constants A only feed the variable-base one)
      const unsigned masks[4] = {BIT(PI_X), BIT(PI_A), BIT(PO_XC), BIT(PO_TH)};
      RC((run_side<C, F1>(c, ".g1", N, sp, arrs, (const S*)pool, pm.total, (const A1*)c->tabs->tab16_g1.p, outs,
                          ov ? c->side[1] : nullptr, c->sev[1], c->sev[2], pair_reds ? &red1 : nullptr, masks)));
      (void)first;
      return GS_OK;
    };
    // G2 side: ycoms (n) + pi (kx).  constants B (len m) multiply R; Psi multiplies Y; fixed part Omega.
    auto side_g2 = [&](bool first) -> int {
      SidePlan sp;
      sp.tm = yg ? pick_tm(c, fillN(c, N), m + n, kx, true, kx) : 1;
      build_side(sp, ycoms != nullptr, n, m, yg, ky, kx, pm.SC, pm.YC, pm.RC, pm.PSI, pm.OM, pm.RHO);
      ArrTab arrs;
      memset(&arrs, 0, sizeof arrs);
      if (yg) {
        arrs.base[0] = (const uint8_t*)Y;
        arrs.stride[0] = shared ? 0u : (uint32_t)(n * Z::G2);
        arrs.base[1] = (const uint8_t*)B;
        arrs.stride[1] = (uint32_t)(m * Z::G2);
      }
      OutTab outs;
      memset(&outs, 0, sizeof outs);
      outs.base[0] = (uint8_t*)ycoms;
      outs.stride[0] = (uint32_t)(n * Z::COM2);
      outs.base[1] = (uint8_t*)pi;
      outs.stride[1] = (uint32_t)(kx * Z::COM2);
      if (ov) c->cur = c->side[0];  // the whole G2 side runs beside the G1 side
      RangeGuard rg("gs.prove.g2");
      const unsigned masks[4] = {BIT(PI_Y), BIT(PI_B), BIT(PO_YC), BIT(PO_PI)};
      RC((run_side<C, F2>(c, ".g2", N, sp, arrs, (const S*)pool, pm.total, (const A2*)c->tabs->tab16_g2.p, outs,
                          ov ? c->side[2] : nullptr, c->sev[3], c->sev[4], pair_reds ? &red2 : nullptr, masks)));
      if (ov) {
        hipEventRecord(c->sev[5], c->side[0]);
        c->cur = nullptr;
        hipStreamWaitEvent(c->stream, c->sev[5], 0);
      }
      (void)first;
      return GS_OK;
    };
    // A host-pointer call runs the G2 side FIRST: every output but the last proof element crosses PCIe under later
    // kernels (the commitments of a side under its variable-base kernel, the first side's proof element under the second
    // side), and the one that is left after the last kernel is then theta (1/8 of the output bytes) rather than pi.
    if (c->pipe && !ov && N >= 16 * c->simd_slots) {
      RC(side_g2(true));
      RC(side_g1(false));
    } else {
      RC(side_g1(true));
      RC(side_g2(false));
    }
    if (pair_reds) {
      hipEventRecord(c->sev[0], c->stream);  // every partial sum of both sides is enqueued before this point
      hipStreamWaitEvent(c->side[0], c->sev[0], 0);
      c->cur = c->side[0];
      RC((run_red<C, F2>(c, N, red2)));
      hipEventRecord(c->sev[5], c->side[0]);
      c->cur = nullptr;
      RC((run_red<C, F1>(c, N, red1)));
      hipStreamWaitEvent(c->stream, c->sev[5], 0);
    }
    return GS_OK;
  }

  // commitments only (commit.rs:59-256): a "prove" side with no proof elements
  template <class F>
  static int commit(gs_ctx* c, size_t count, bool group, const void* vars, const void* rand, void* out,
                    const Aff<F>* tab, const char* tag) {
    int kc = group ? 2 : 1;
    const size_t bsz = sizeof(Aff<F>) == sizeof(A1) ? Z::G1 : Z::G2;  // boundary bytes of one point
    PoolMap pm;
    memset(&pm, 0, sizeof pm);
    pm.RC = 0;
    pm.XC = kc;
    pm.total = kc + 1;
    void* pool;
    RC(scratch(c, "commit.pool", count * pm.total * sizeof(S), &pool));
    // reuse k_prep_prove with m = 1 variable per "equation", no Gamma work (n = 0, ky = 0)
    RC(launch_seg<k_prep_prove<C>>(c, "k_prep_commit", count, 64, count, 1, 0, kc, 0, (const S*)nullptr, (const S*)rand,
              (const S*)nullptr, (const S*)nullptr, group ? nullptr : (const S*)vars, (const S*)nullptr,
              (const S*)nullptr, (const S*)nullptr, pm, (S*)pool, 0));
    SidePlan sp;
    build_side(sp, true, 1, 0, group, kc, 0, pm.RC, pm.XC, 0, 0, 0, 0);
    ArrTab arrs;
    memset(&arrs, 0, sizeof arrs);
    if (group) {
      arrs.base[0] = (const uint8_t*)vars;
      arrs.stride[0] = (uint32_t)bsz;
    }
    OutTab outs;
    memset(&outs, 0, sizeof outs);
    outs.base[0] = (uint8_t*)out;
    outs.stride[0] = (uint32_t)(2 * bsz);
    return run_side<C, F>(c, tag, count, sp, arrs, (const S*)pool, pm.total, tab, outs);
  }

  // --------------------------------------------------------------- verify --
  struct VerifyPlan {
    SidePlan g1;                      // PA_j (j<n) [+ PB] as Com1-shaped pairs
    std::vector<MillerTask> mt;
    CellMap cm;
    int npa;                          // Com1 elements in the PA scratch per equation
  };

  static void add_pair(std::vector<PairRef>& v, int p_arr, int p_idx, int neg, int q_arr, int q_idx) {
    PairRef r;
    r.p_arr = (uint8_t)p_arr;
    r.p_idx = (uint32_t)p_idx;
    r.neg = (uint8_t)neg;
    r.q_arr = (uint8_t)q_arr;
    r.q_idx = (uint32_t)q_idx;
    r.pad = 0;
    v.push_back(r);
  }

  // P arrays: 0 PA scratch, 1 xcoms, 2 crs G1 consts, 3 theta
  // Q arrays: 0 ycoms, 1 B, 2 crs G2 consts, 3 pi, 4 target (MSMEG2)
  static void build_verify(VerifyPlan& vp, int curve, int ty, int m, int n, const PoolMap& pm, double budget, bool twin,
                           int tm, bool lt) {
    build_verify_g1(vp, ty, m, n, pm, tm);
    build_verify_miller(vp, curve, ty, m, n, budget, twin, lt);
  }
  static void build_verify_g1(VerifyPlan& vp, int ty, int m, int n, const PoolMap& pm, int tm) {
    bool xg = x_is_group(ty), yg = y_is_group(ty);
    // ---- G1-side points: PA_j.a = map_a_j.a + sum_i Gamma_ij c_i.a ;  PB.a = sum_i b_i c_i.a - lin_t
    // engine arrays: 0 = xcoms as 2m G1 points, 1 = A (group), 2 = target (MSMEG1)
    SidePlan& sp = vp.g1;
    sp.tm = tm;
    if (tm < 0) {  // shared per-base tables: groups of 8 terms, array 0 = the 2 m components of the X commitments
      sp.tm = 8;
      sp.tab_bases = 2 * m;
    }
    int slot = 0;
    for (int j = 0; j < n; j++) {
      int b[2], e[2];
      for (int a = 0; a < 2; a++) {
        b[a] = slot;
        if (xg) {
          if (a == 1) sp.fix.push_back(mkfix(0, 0xFF, 0, 0xFF, 1, j, slot++));
        } else {
          sp.fix.push_back(mkfix(pm.AC + j, tb_w(a), 0, 0xFF, 0xFF, 0, slot++));
        }
        {
          std::vector<VarTask> terms;
          for (int i = 0; i < m; i++) terms.push_back(mkvar(pm.GC + i * n + j, 0, 2 * i + a, 0));
          add_var_terms(sp, terms, slot);
        }
        e[a] = slot;
      }
      sp.red.push_back(mkred(b[0], e[0], b[1], e[1], 0, j));
    }
    vp.npa = n;
    if (!yg) {
      int b[2], e[2];
      for (int a = 0; a < 2; a++) {
        b[a] = slot;
        {
          std::vector<VarTask> terms;
          for (int i = 0; i < m; i++) terms.push_back(mkvar(pm.BC + i, 0, 2 * i + a, 0));
          add_var_terms(sp, terms, slot);
        }
        if (ty == GS_MSMEG1 && a == 1) sp.fix.push_back(mkfix(0, 0xFF, 0, 0xFF, 2, 0, slot++, 1));  // - t
        if (ty == GS_QUAD) sp.fix.push_back(mkfix(pm.NT, tb_w(a), 0, 0xFF, 0xFF, 0, slot++));       // (-t) W1.a
        e[a] = slot;
      }
      sp.red.push_back(mkred(b[0], e[0], b[1], e[1], 0, n));
      vp.npa = n + 1;
    }
    sp.nslots = slot;
  }
  static void build_verify_miller(VerifyPlan& vp, int curve, int ty, int m, int n, double budget, bool twin, bool lt) {
    bool xg = x_is_group(ty), yg = y_is_group(ty);
    int kx = xg ? 2 : 1, ky = yg ? 2 : 1;
    // ---- Miller tasks.  Pairs of cell (a, b): G1 argument = component a, G2 argument = component b.
    //  twin  : one task list per b, each lane takes `ch` (Q, P0, P1) triples and keeps two accumulators
    //          (lines of Q computed once) -- less total work, pays off once the chip is full;
    //  single: one task list per cell, `ch` (P, Q) pairs and one accumulator per lane -- more, shorter lanes.
    vp.mt.clear();
    for (int cell = 0; cell < (twin ? 2 : 4); cell++) {
      int b = twin ? cell : (cell & 1), a = twin ? 0 : (cell >> 1);
      std::vector<PairRef> pr;
      for (int j = 0; j < n; j++) add_pair(pr, 0, 2 * j + a, 0, 0, 2 * j + b);      // (PA_j.a, d_j.b)
      if (yg) {
        if (b == 1)
          for (int i = 0; i < m; i++) add_pair(pr, 1, 2 * i + a, 0, 1, i);          // (c_i.a, B_i)
      } else {
        add_pair(pr, 0, 2 * n + a, 0, 2, 4 + b);                                    // (PB.a, W2.b)
      }
      for (int k = 0; k < kx; k++) add_pair(pr, 2, 2 * k + a, 1, 3, 2 * k + b);     // (-u_k.a, pi_k.b)
      for (int l = 0; l < ky; l++) add_pair(pr, 3, 2 * l + a, 1, 2, 2 * l + b);     // (-theta_l.a, v_l.b)
      if (ty == GS_MSMEG2 && b == 1) add_pair(pr, 2, 4 + a, 1, 4, 0);               // (-W1.a, t)
      int lo = (int)vp.mt.size();
      chunk_tasks(vp.mt, pr, budget, b, !twin, mcost(curve, twin), lt);
      int hi = (int)vp.mt.size();
      if (twin) {
        for (int aa = 0; aa < 2; aa++) {
          vp.cm.lo[2 * aa + b] = lo;
          vp.cm.hi[2 * aa + b] = hi;
          vp.cm.sub[2 * aa + b] = aa;
        }
      } else {
        vp.cm.lo[cell] = lo;
        vp.cm.hi[cell] = hi;
        vp.cm.sub[cell] = 0;
      }
    }
  }

  // shared front of both verifier modes: G1-side points + Miller partials
  static int verify_front(gs_ctx* c, int ty, size_t N, int m, int n, const void* A, const void* B, const void* G,
                          const void* target, const void* xcoms, const void* ycoms, const void* pi,
                          const void* theta, VerifyPlan& vp, void** mpart_out, bool shared = false) {
    bool xg = x_is_group(ty), yg = y_is_group(ty);
    int kx = xg ? 2 : 1, ky = yg ? 2 : 1;
    PoolMap pm;
    memset(&pm, 0, sizeof pm);
    int o = 0;
    pm.GC = o; o += m * n;
    pm.AC = o; o += n;
    pm.BC = o; o += m;
    pm.NT = o; o += 1;
    pm.total = o;
    void* pool;
    RC(scratch(c, "verify.pool", N * pm.total * sizeof(S), &pool));
    RC(need(c, BIT(VI_G) | (xg ? 0u : BIT(VI_A)) | (yg ? 0u : BIT(VI_B)) | (ty == GS_QUAD ? BIT(VI_TG) : 0u)));
    const bool wide = wide_prep(m, n);
    if (wide)
      RC(launch_seg<k_fr_canonical<C>>(c, "k_fr_canonical", N * (size_t)m * n, 64, N * (size_t)m * n, m * n, (const S*)G,
                pm.total, (S*)pool + pm.GC));
    RC(launch_seg<k_prep_verify<C>>(c, "k_prep_verify", N, 64, N, m, n, wide ? nullptr : (const S*)G,
              xg ? nullptr : (const S*)A, yg ? nullptr : (const S*)B, ty == GS_QUAD ? (const S*)target : nullptr, pm,
              (S*)pool));
    // lane shape for this batch size: single or twin accumulators (the twin task list on one lane with two
    // accumulators, or on a pair of lanes with one each), lane-cost budget (cost model above)
    int mode = 0;  // 0 single, 1 twin, 2 pair (LDS exchange), 3 pair (DPP exchange)
    double budget = 3 * 4621.0;
    {
      // the choice depends on the shape, the batch size and the overrides only: remembered per context
      char key[96];
      const size_t NF = fillN(c, N);
      snprintf(key, sizeof key, "%d.%d.%d.%zu.%d.%d.%d", ty, m, n, NF, c->miller_twin, c->miller_ch, (int)c->line_tables);
      auto hit = c->miller_choice.find(key);
      if (hit != c->miller_choice.end()) {
        mode = hit->second.first;
        budget = hit->second.second;
      } else {
        double best = -1;
        for (int md = 0; md < 3; md++) {
          if (c->miller_twin >= 0 && md != (c->miller_twin == 3 ? 2 : c->miller_twin)) continue;
          if (c->miller_twin < 0 && md == 2 && !GS_PLAN_PAIR) continue;
          for (double cand : miller_budgets(c, md != 0)) {
            VerifyPlan tmp;
            build_verify_miller(tmp, c->curve, ty, m, n, cand, md != 0, c->line_tables);
            double cost = miller_cost(c, NF, tmp.mt, md != 0, md == 2);
            if (md == 2) cost *= 0.97;  // measured: the same triples finish 3-4 % sooner with one accumulator per lane
            if (best < 0 || cost < best) {
              best = cost;
              mode = md;
              budget = cand;
            }
          }
        }
        // planned exchange of the pair kernel: LDS slots on BLS12-381, DPP on BN254.  (DPP won until the round loop's
        // products moved into one lambda, 166.7 vs 170.0 ms; with that shape the partner's line parked in LDS across
        // the lane's own product beats the 84 registers it occupies: 147.6 vs 155.0 ms at 2^16, 41.3 vs 42.5 at 2^14,
        // 12.7 vs 13.2 at 2^12; BN254's shorter lines keep DPP ahead by 0.6 %: profiles/r3/ab_exchange.txt)
        if (mode == 2) mode = c->miller_twin == 3 ? 3 : c->miller_twin == 2 ? 2 : (c->curve == 0 ? 2 : 3);
        if (c->miller_choice.size() > 4096) c->miller_choice.clear();
        c->miller_choice[key] = std::make_pair(mode, budget);
      }
    }
    const bool twin = mode != 0;
    // the Gamma^T c lanes: Straus groups with their own tables, or (large arities: every base serves 2 n outputs)
    // lanes over shared per-base window tables -- measured in profiles/r3/large_arity_334.json
    // (planned only while the tables -- N x 2 m bases x 128 entries, affine + Jacobian staging -- stay below 8 GB)
    const double tab8_bytes = (double)N * 2.0 * m * 128.0 * (double)(sizeof(A1) + sizeof(Jac<F1>));
    // (the shared base tables are read by endomorphism-decomposed lanes: not with "endo" off)
    const bool tab8 = c->endo && (c->var_tab == 1 || (c->var_tab < 0 && wide_prep(m, n) && n >= 32 && tab8_bytes <= 8e9));
    build_verify(vp, c->curve, ty, m, n, pm, budget, twin, tab8 ? -1 : pick_tm(c, fillN(c, N), m, 2 * n, false, n),
                 c->line_tables);
    // G1-side points
    void* pa;
    RC(scratch(c, "verify.pa", N * vp.npa * Z::COM1, &pa));
    {
      ArrTab arrs;
      memset(&arrs, 0, sizeof arrs);
      arrs.base[0] = (const uint8_t*)xcoms;
      arrs.stride[0] = shared ? 0u : (uint32_t)(m * Z::COM1);  // a Statement's equations share the commitments
      if (xg) {
        arrs.base[1] = (const uint8_t*)A;
        arrs.stride[1] = (uint32_t)(n * Z::G1);
      }
      if (ty == GS_MSMEG1) {
        arrs.base[2] = (const uint8_t*)target;
        arrs.stride[2] = (uint32_t)Z::G1;
      }
      OutTab outs;
      memset(&outs, 0, sizeof outs);
      outs.base[0] = (uint8_t*)pa;
      outs.stride[0] = (uint32_t)(vp.npa * Z::COM1);
      RangeGuard rg("gs.verify.g1");
      RC(need(c, BIT(VI_XC) | BIT(VI_A) | BIT(VI_TG) * (ty == GS_MSMEG1 ? 1u : 0u)));
      RC((run_side<C, F1>(c, ".vg1", N, vp.g1, arrs, (const S*)pool, pm.total, (const A1*)c->tabs->tab16_g1.p, outs)));
    }
    // Miller
    RangeGuard rg_m("gs.verify.miller");
    RC(need(c, 0xFFu));  // every input array from here on
    const MillerTask* dmt;
    RC(upload(c, "verify.mt", vp.mt, &dmt));
    int ntask = (int)vp.mt.size();
    void* mpart;
    RC(scratch(c, "verify.mpart", 2 * N * ntask * sizeof(GT), &mpart));
    ArrTab parr, qarr;
    memset(&parr, 0, sizeof parr);
    memset(&qarr, 0, sizeof qarr);
    parr.base[0] = (const uint8_t*)pa;
    parr.stride[0] = (uint32_t)(vp.npa * Z::COM1);
    parr.base[1] = (const uint8_t*)xcoms;
    parr.stride[1] = shared ? 0u : (uint32_t)(m * Z::COM1);
    parr.base[2] = (const uint8_t*)c->tabs->crs_g1.p;
    parr.stride[2] = 0;
    parr.base[3] = (const uint8_t*)theta;
    parr.stride[3] = (uint32_t)(ky * Z::COM1);
    qarr.base[0] = (const uint8_t*)ycoms;
    qarr.stride[0] = shared ? 0u : (uint32_t)(n * Z::COM2);
    qarr.base[1] = (const uint8_t*)B;
    qarr.stride[1] = (uint32_t)(m * Z::G2);
    qarr.base[2] = (const uint8_t*)c->tabs->crs_g2.p;
    qarr.stride[2] = 0;
    qarr.base[3] = (const uint8_t*)pi;
    qarr.stride[3] = (uint32_t)(kx * Z::COM2);
    qarr.base[4] = (const uint8_t*)target;
    qarr.stride[4] = (uint32_t)Z::G2;
    {
      // (P, Q) pairs (twin: (Q, P0, P1) triples) per equation, a table-reading pair counted at its cost ratio
      const MCost mc = mcost(c->curve, twin);
      double pairs = 0;
      for (const MillerTask& t : vp.mt)
        for (int q = 0; q < t.np; q++) pairs += pair_fixed(c->line_tables, t.pr[q]) ? mc.fix / mc.var : 1.0;
      c->work_hint = (uint64_t)((double)N * pairs);
    }
    if (getenv("GS_PLAN_TRACE")) {
      const MCost mc = mcost(c->curve, twin);
      fprintf(stderr, "[plan] verify ty %d N %zu (fill %zu) m %d n %d: miller mode %d budget %.0f, %d tasks, lanes:", ty, N,
              fillN(c, N), m, n, mode, budget, (int)ntask);
      for (const MillerTask& t : vp.mt) {
        double l = mc.base;
        for (int q = 0; q < t.np; q++) l += pair_fixed(c->line_tables, t.pr[q]) ? mc.fix : mc.var;
        fprintf(stderr, " %.0f", mode >= 2 ? pair_lane_cost(c, t) : l);
      }
      fprintf(stderr, "; waves %zu\n", ((mode >= 2 ? 2 : 1) * N * ntask + 63) / 64);
    }
    if (mode == 2)
      RC(launch_seg<k_miller_pair<C, false>>(c, "k_miller.pair", 2 * N * ntask, 64, 2 * N * ntask, ntask, dmt, parr, qarr,
                (GT*)mpart, (const Line<C>*)(c->line_tables ? c->tabs->line_tab.p : nullptr)));
    else if (mode == 3)
      RC(launch_seg<k_miller_pair<C, true>>(c, "k_miller.pairdpp", 2 * N * ntask, 64, 2 * N * ntask, ntask, dmt, parr, qarr,
                (GT*)mpart, (const Line<C>*)(c->line_tables ? c->tabs->line_tab.p : nullptr)));
    else if (twin)
      RC(launch_seg<k_miller<C, true>>(c, "k_miller.twin", N * ntask, 64, N * ntask, ntask, dmt, parr, qarr, (GT*)mpart, 2,
                (const Line<C>*)(c->line_tables ? c->tabs->line_tab.p : nullptr)));
    else
      RC(launch_seg<k_miller<C, false>>(c, "k_miller", N * ntask, 64, N * ntask, ntask, dmt, parr, qarr, (GT*)mpart, 2,
                (const Line<C>*)(c->line_tables ? c->tabs->line_tab.p : nullptr)));
    *mpart_out = mpart;
    return GS_OK;
  }

  static int verify(gs_ctx* c, int ty, size_t N, int m, int n, const void* A, const void* B, const void* G,
                    const void* target, const void* xcoms, const void* ycoms, const void* pi, const void* theta,
                    uint8_t* ok, bool shared = false) {
    VerifyPlan vp;
    void* mpart;
    RangeGuard rg_all("gs.verify");
    RC(verify_front(c, ty, N, m, n, A, B, G, target, xcoms, ycoms, pi, theta, vp, &mpart, shared));
    RangeGuard rg_f("gs.verify.final");
    int ntask = (int)vp.mt.size();
    // large arities: hundreds of Miller partials per cell -- fold runs of K in parallel until k_final's own serial
    // product is short (segmented K-ary tree in GT)
    CellMap cm = vp.cm;
    {
      const int K = 8;
      for (int level = 0;; level++) {
        int longest = 0;
        for (int q = 0; q < 4; q++) longest = std::max(longest, cm.hi[q] - cm.lo[q]);
        if (longest <= 2 * K) break;
        CellMap out;
        int nt_out = 0, runs_max = (longest + K - 1) / K;
        for (int q = 0; q < 4; q++) {
          out.lo[q] = nt_out;
          nt_out += (cm.hi[q] - cm.lo[q] + K - 1) / K;
          out.hi[q] = nt_out;
          out.sub[q] = 0;
        }
        void* folded;
        RC(scratch(c, level & 1 ? "verify.mfold1" : "verify.mfold0", 2 * N * (size_t)nt_out * sizeof(GT), &folded));
        size_t total = N * 4 * (size_t)runs_max;
        RC(launch_seg<k_cell_fold<C>>(c, "k_cell_fold", total, 64, total, runs_max, ntask, cm, (const GT*)mpart, K, nt_out, out,
                  (GT*)folded));
        mpart = folded;
        ntask = nt_out;
        cm = out;
      }
    }
    void* cellok;
    RC(scratch(c, "verify.cellok", N * 4, &cellok));
    // one lane per final exponentiation once that fills the chip, 3-lane groups (21 per wave) below that
    size_t coop_waves = (N * 4 + 20) / 21;
    if (c->coop_fe == 2 || (c->coop_fe == 1 && (fillN(c, N) * 4 + 20) / 21 <= c->simd_slots))
      RC(launch_seg<k_final_coop<C>>(c, "k_final.coop", coop_waves * 63, 63, N, ntask, cm, (const GT*)mpart,
                ty == GS_PPE ? (const uint8_t*)target : nullptr, (uint8_t*)cellok));
    else
      RC(launch_seg<k_final<C>>(c, "k_final", N * 4, 64, N, ntask, cm,
                (const GT*)mpart, ty == GS_PPE ? (const uint8_t*)target : nullptr, (uint8_t*)cellok));
    RC(launch_seg<k_and4>(c, "k_and4", N, 64, N, (const uint8_t*)cellok, ok));
    return GS_OK;
  }

  // product of n internal-form GT elements at `buf` (clobbers buf/tmp); result exported in
  // boundary form to dst (one GT)
  static int gt_product(gs_ctx* c, size_t n, GT* buf, GT* tmp, uint8_t* dst) {
    const int K = 8;
    GT *in = buf, *out = tmp;
    while (n > 1) {
      size_t no = (n + K - 1) / K;
      RC(launch(c, "k_gt_prod", k_gt_prod<C>, no, 64, n, (const GT*)in, no, out, K));
      GT* t = in;
      in = out;
      out = t;
      n = no;
    }
    RC(launch(c, "k_gt_export", k_gt_export<C>, 1, 64, (size_t)1, (const GT*)in, dst));
    return GS_OK;
  }

  // Batched verifier.  acc[0] = prod_e prod_cells Miller(e,cell)^rho_{e,cell} (un-exponentiated),
  // acc[1] = prod_e t_e^rho_{e,11} (1 for non-PPE).  The exponents are FOLDED INTO THE G1 ARGUMENTS:
  //   prod_{a,b} e(P_a, Q_b)^{rho_ab} = prod_b e(rho_0b P_0 + rho_1b P_1, Q_b)
  // so every G2 argument is paired ONCE (20 instead of 40 pairs for the 4+4 PPE), there is one
  // accumulator per Miller lane, no per-cell exponentiation and one final exponentiation per batch.
  // Two passes of the linear-combination engine on G1:
  //   1. C'_ib = rho_0b c_i.0 + rho_1b c_i.1,  Th'_lb = rho_0b theta_l.0 + rho_1b theta_l.1
  //   2. P'_jb = rho_1b A_j + sum_i Gamma_ij C'_ib   (a_j rho_ab W1.a for scalar constants),
  //      PB'_b, U'_kb = rho_0b u_k.0 + rho_1b u_k.1, W1'_b      (Com-shaped: components b = 0, 1)
  static int verify_rlc(gs_ctx* c, int ty, size_t N, int m, int n, const void* A, const void* B, const void* G,
                        const void* target, const void* xcoms, const void* ycoms, const void* pi, const void* theta,
                        const uint64_t* rho, void* acc) {
    bool xg = x_is_group(ty), yg = y_is_group(ty);
    int kx = xg ? 2 : 1, ky = yg ? 2 : 1;
    PoolMap pm;
    memset(&pm, 0, sizeof pm);
    int o = 0;
    pm.GC = o; o += m * n;
    pm.BC = o; o += m;
    pm.RH = o; o += 4;
    pm.AR = o; o += 4 * n;
    pm.NR = o; o += 4;
    pm.total = o;
    void* pool;
    RC(scratch(c, "rlc.pool", N * pm.total * sizeof(S), &pool));
    const bool wide = wide_prep(m, n);
    if (wide)
      RC(launch_seg<k_fr_canonical<C>>(c, "k_fr_canonical", N * (size_t)m * n, 64, N * (size_t)m * n, m * n, (const S*)G,
                pm.total, (S*)pool + pm.GC));
    RC(launch(c, "k_prep_verify_rlc", k_prep_verify_rlc<C>, N, 64, N, m, n, wide ? nullptr : (const S*)G,
              xg ? nullptr : (const S*)A, yg ? nullptr : (const S*)B, ty == GS_QUAD ? (const S*)target : nullptr, rho,
              pm, (S*)pool));
    auto RH = [&](int a, int b) { return pm.RH + 2 * a + b; };
    // ---- pass 1
    int n1 = m + ky;
    void* s1;
    RC(scratch(c, "rlc.s1", N * n1 * Z::COM1, &s1));
    {
      SidePlan sp;
      sp.tm = 4;
      int slot = 0;
      for (int q = 0; q < n1; q++) {
        int arr = q < m ? 0 : 1, idx = q < m ? q : q - m;
        int b0 = slot, e0 = 0, b1 = 0;
        for (int b = 0; b < 2; b++) {
          if (b == 1) { e0 = slot; b1 = slot; }
          std::vector<VarTask> terms;
          terms.push_back(mkvar(RH(0, b), arr, 2 * idx, 0));
          terms.push_back(mkvar(RH(1, b), arr, 2 * idx + 1, 0));
          add_var_terms(sp, terms, slot);
        }
        sp.red.push_back(mkred(b0, e0, b1, slot, 0, q));
      }
      sp.nslots = slot;
      ArrTab arrs;
      memset(&arrs, 0, sizeof arrs);
      arrs.base[0] = (const uint8_t*)xcoms;
      arrs.stride[0] = (uint32_t)(m * Z::COM1);
      arrs.base[1] = (const uint8_t*)theta;
      arrs.stride[1] = (uint32_t)(ky * Z::COM1);
      OutTab outs;
      memset(&outs, 0, sizeof outs);
      outs.base[0] = (uint8_t*)s1;
      outs.stride[0] = (uint32_t)(n1 * Z::COM1);
      RC((run_side<C, F1>(c, ".r1", N, sp, arrs, (const S*)pool, pm.total, (const A1*)c->tabs->tab16_g1.p, outs)));
    }
    // ---- pass 2: elements 0..n-1 = P'_j, then [PB'] , U'_k (kx), [W1'] (MSMEG2)
    int iPB = n, iU = n + (yg ? 0 : 1), iW = iU + kx, n2 = iW + (ty == GS_MSMEG2 ? 1 : 0);
    void* s2;
    RC(scratch(c, "rlc.s2", N * n2 * Z::COM1, &s2));
    {
      SidePlan sp;
      sp.tm = 8;
      int slot = 0;
      for (int q = 0; q < n2; q++) {
        int b0 = slot, e0 = 0, b1 = 0;
        for (int b = 0; b < 2; b++) {
          if (b == 1) { e0 = slot; b1 = slot; }
          std::vector<VarTask> terms;
          if (q < n) {
            int j = q;
            if (xg)
              terms.push_back(mkvar(RH(1, b), 1, j, 0));                                        // rho_1b A_j
            else
              sp.fix.push_back(mkfix(pm.AR + j * 4 + 0 + b, tb_w(0), pm.AR + j * 4 + 2 + b, tb_w(1), 0xFF, 0, slot++));
            for (int i = 0; i < m; i++) terms.push_back(mkvar(pm.GC + i * n + j, 0, 2 * i + b, 0));  // Gamma_ij C'_ib
          } else if (!yg && q == iPB) {
            for (int i = 0; i < m; i++) terms.push_back(mkvar(pm.BC + i, 0, 2 * i + b, 0));          // b_i C'_ib
            if (ty == GS_MSMEG1) {
              VarTask v = mkvar(RH(1, b), 2, 0, 0);                                                  // - rho_1b t
              v.neg = 1;
              terms.push_back(v);
            }
            if (ty == GS_QUAD) sp.fix.push_back(mkfix(pm.NR + 0 + b, tb_w(0), pm.NR + 2 + b, tb_w(1), 0xFF, 0, slot++));
          } else if (q >= iU && q < iU + kx) {
            int k = q - iU;
            sp.fix.push_back(mkfix(RH(0, b), tb_u(k, 0), RH(1, b), tb_u(k, 1), 0xFF, 0, slot++));
          } else {  // W1' (MSMEG2)
            sp.fix.push_back(mkfix(RH(0, b), tb_w(0), RH(1, b), tb_w(1), 0xFF, 0, slot++));
          }
          if (!terms.empty()) add_var_terms(sp, terms, slot);
        }
        sp.red.push_back(mkred(b0, e0, b1, slot, 0, q));
      }
      sp.nslots = slot;
      ArrTab arrs;
      memset(&arrs, 0, sizeof arrs);
      arrs.base[0] = (const uint8_t*)s1;
      arrs.stride[0] = (uint32_t)(n1 * Z::COM1);
      if (xg) {
        arrs.base[1] = (const uint8_t*)A;
        arrs.stride[1] = (uint32_t)(n * Z::G1);
      }
      if (ty == GS_MSMEG1) {
        arrs.base[2] = (const uint8_t*)target;
        arrs.stride[2] = (uint32_t)Z::G1;
      }
      OutTab outs;
      memset(&outs, 0, sizeof outs);
      outs.base[0] = (uint8_t*)s2;
      outs.stride[0] = (uint32_t)(n2 * Z::COM1);
      RC((run_side<C, F1>(c, ".r2", N, sp, arrs, (const S*)pool, pm.total, (const A1*)c->tabs->tab16_g1.p, outs)));
    }
    // ---- Miller: every G2 argument once.  P arrays: 0 = S2, 1 = S1.  Q arrays: 0 ycoms, 1 B, 2 crs, 3 pi, 4 target
    std::vector<PairRef> pr;
    for (int b = 0; b < 2; b++) {
      for (int j = 0; j < n; j++) add_pair(pr, 0, 2 * j + b, 0, 0, 2 * j + b);
      if (yg) {
        if (b == 1)
          for (int i = 0; i < m; i++) add_pair(pr, 1, 2 * i + 1, 0, 1, i);
      } else {
        add_pair(pr, 0, 2 * iPB + b, 0, 2, 4 + b);
      }
      for (int k = 0; k < kx; k++) add_pair(pr, 0, 2 * (iU + k) + b, 1, 3, 2 * k + b);
      for (int l = 0; l < ky; l++) add_pair(pr, 1, 2 * (m + l) + b, 1, 2, 2 * l + b);
      if (ty == GS_MSMEG2 && b == 1) add_pair(pr, 0, 2 * iW + 1, 1, 4, 0);
    }
    std::vector<MillerTask> mt;
    {
      double best = -1;
      for (double cand : miller_budgets(c, false)) {
        std::vector<MillerTask> tmp;
        chunk_tasks(tmp, pr, cand, 0, true, mcost(c->curve, false), c->line_tables);
        double cost = miller_cost(c, N, tmp, false);
        if (best < 0 || cost < best) {
          best = cost;
          mt = tmp;
        }
      }
    }
    const MillerTask* dmt;
    RC(upload(c, "rlc.mt", mt, &dmt));
    int ntask = (int)mt.size();
    void *mpart, *pt, *tmp;
    RC(scratch(c, "rlc.mpart", N * ntask * sizeof(GT), &mpart));
    RC(scratch(c, "rlc.t", (N + 1) * sizeof(GT), &pt));
    RC(scratch(c, "rlc.tmp", (N * ntask / 4 + 8) * sizeof(GT), &tmp));
    ArrTab parr, qarr;
    memset(&parr, 0, sizeof parr);
    memset(&qarr, 0, sizeof qarr);
    parr.base[0] = (const uint8_t*)s2;
    parr.stride[0] = (uint32_t)(n2 * Z::COM1);
    parr.base[1] = (const uint8_t*)s1;
    parr.stride[1] = (uint32_t)(n1 * Z::COM1);
    qarr.base[0] = (const uint8_t*)ycoms;
    qarr.stride[0] = (uint32_t)(n * Z::COM2);
    qarr.base[1] = (const uint8_t*)B;
    qarr.stride[1] = (uint32_t)(m * Z::G2);
    qarr.base[2] = (const uint8_t*)c->tabs->crs_g2.p;
    qarr.base[3] = (const uint8_t*)pi;
    qarr.stride[3] = (uint32_t)(kx * Z::COM2);
    qarr.base[4] = (const uint8_t*)target;
    qarr.stride[4] = (uint32_t)Z::G2;
    {
      const MCost mc = mcost(c->curve, false);
      double pairs = 0;
      for (const MillerTask& t : mt)
        for (int q = 0; q < t.np; q++) pairs += pair_fixed(c->line_tables, t.pr[q]) ? mc.fix / mc.var : 1.0;
      c->work_hint = (uint64_t)((double)N * pairs);
    }
    RC(launch_seg<k_miller<C, false>>(c, "k_miller.rlc", N * ntask, 64, N * ntask, ntask, dmt, parr, qarr, (GT*)mpart, 1,
              (const Line<C>*)(c->line_tables ? c->tabs->line_tab.p : nullptr)));
    uint8_t* a = (uint8_t*)acc;
    RC(gt_product(c, N * ntask, (GT*)mpart, (GT*)tmp, a));
    if (ty == GS_PPE) {
      RC(launch(c, "k_rlc_tpow", k_rlc_tpow<C>, N, 64, N, (const uint8_t*)target, rho, (GT*)pt));
      RC(gt_product(c, N, (GT*)pt, (GT*)tmp, a + Z::GT));
    } else {
      RC(launch(c, "k_gt_set_one", k_gt_set_one<C>, 1, 64, (GT*)pt));
      RC(launch(c, "k_gt_export", k_gt_export<C>, 1, 64, (size_t)1, (const GT*)pt, a + Z::GT));
    }
    return GS_OK;
  }

  // accs: count boundary-form pairs (host).  ok = FE(prod accs[i][0]) == prod accs[i][1]
  // accs: `count` interleaved pairs (f_i, t_i) in boundary form, on the host or (dev) on this context's device
  static int gt_finalize(gs_ctx* c, size_t count, const void* accs_host, uint8_t* ok_host, bool dev = false) {
    void *raw, *d, *t, *two, *dok, *twob;
    RC(scratch(c, "fin.raw", 2 * count * Z::GT, &raw));
    RC(scratch(c, "fin.in", 2 * count * sizeof(GT), &d));
    RC(scratch(c, "fin.tmp", (count + 8) * sizeof(GT), &t));
    RC(scratch(c, "fin.two", 2 * sizeof(GT), &two));
    RC(scratch(c, "fin.twob", 2 * Z::GT, &twob));
    RC(scratch(c, "fin.ok", 16, &dok));
    // de-interleave: [f0 f1 ...][t0 t1 ...]
    const uint8_t* src = (const uint8_t*)accs_host;
    if (dev) {  // two strided device-to-device copies on the context's stream
      HIPCHK(c, hipMemcpy2DAsync(raw, Z::GT, src, 2 * Z::GT, Z::GT, count, hipMemcpyDeviceToDevice, c->stream));
      HIPCHK(c, hipMemcpy2DAsync((uint8_t*)raw + count * Z::GT, Z::GT, src + Z::GT, 2 * Z::GT, Z::GT, count,
                                 hipMemcpyDeviceToDevice, c->stream));
    } else {
      std::vector<uint8_t> h(2 * count * Z::GT);
      for (size_t i = 0; i < count; i++) {
        memcpy(&h[i * Z::GT], src + (2 * i) * Z::GT, Z::GT);
        memcpy(&h[(count + i) * Z::GT], src + (2 * i + 1) * Z::GT, Z::GT);
      }
      HIPCHK(c, hipStreamSynchronize(c->stream));
      HIPCHK(c, hipMemcpy(raw, h.data(), h.size(), hipMemcpyHostToDevice));
    }
    RC(launch(c, "k_gt_import", k_gt_import<C>, 2 * count, 64, 2 * count, (const uint8_t*)raw, (GT*)d));
    RC(gt_product(c, count, (GT*)d, (GT*)t, (uint8_t*)twob));
    RC(gt_product(c, count, (GT*)d + count, (GT*)t, (uint8_t*)twob + Z::GT));
    RC(launch(c, "k_gt_import", k_gt_import<C>, 2, 64, (size_t)2, (const uint8_t*)twob, (GT*)two));
    RC(launch(c, "k_fe_eq", k_fe_eq<C>, 3, 3, (const GT*)two, (uint8_t*)dok));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipMemcpy(ok_host, dok, 1, hipMemcpyDeviceToHost));
    return GS_OK;
  }
};

// ---------------------------------------------------------------------------
// dispatch helpers
// ---------------------------------------------------------------------------
#if defined(GS_ONLY_BLS)
// experiment builds (tools/build_variant.sh NAME -DGS_ONLY_BLS): half the compile time, BN254 calls fail
#define DISPATCH(ctx, EXPR) \
  ((ctx)->curve == GS_CURVE_BLS12_381 ? Impl<Bls12_381>::EXPR : fail(ctx, GS_ERR_ARG, "built with GS_ONLY_BLS"))
#else
#define DISPATCH(ctx, EXPR)                                   \
  ((ctx)->curve == GS_CURVE_BLS12_381 ? Impl<Bls12_381>::EXPR : Impl<Bn254>::EXPR)
#endif

#if defined(GS_ONLY_BLS)
#define WIRE_DISPATCH(expr_bls, expr_bn) (c->curve == 0 ? (expr_bls) : fail(c, GS_ERR_ARG, "built with GS_ONLY_BLS"))
#else
#define WIRE_DISPATCH(expr_bls, expr_bn) (c->curve == 0 ? (expr_bls) : (expr_bn))
#endif

static size_t sz_fq(int curve) { return curve == 0 ? 4 * Bls12_381::N : 4 * Bn254::N; }
static const size_t SZ_FR = 32;

struct HostStage {  // host<->device staging for the un-suffixed entry points
  // Buffers come from the context's grow-only scratch map ("stage.<k>"): no hipMalloc / hipFree in steady state
  // (hipFree synchronises the device; per-call allocation cost 10-20 % of a 2^12 batch).
  gs_ctx* c;
  int k = 0;
  explicit HostStage(gs_ctx* ctx) : c(ctx) {}
  int slot(size_t bytes, void** d) {
    char name[32];
    snprintf(name, sizeof name, "stage.%d", k++);
    return scratch(c, name, bytes, d);
  }
  int in(const void* h, size_t bytes, void** d) {
    *d = nullptr;
    if (!h || bytes == 0) return GS_OK;
    RC(slot(bytes, d));
    hipError_t e = hipMemcpyAsync(*d, h, bytes, hipMemcpyHostToDevice, c->stream);
    if (e != hipSuccess) return fail(c, GS_ERR_DEVICE, "hipMemcpy H2D", e);
    return GS_OK;
  }
  int out(void* h, size_t bytes, void** d) {
    *d = nullptr;
    if (!h || bytes == 0) return GS_OK;
    return slot(bytes, d);
  }
  int back(void* h, const void* d, size_t bytes) {
    if (!h || bytes == 0) return GS_OK;
    hipError_t e = hipMemcpyAsync(h, d, bytes, hipMemcpyDeviceToHost, c->stream);
    if (e != hipSuccess) return fail(c, GS_ERR_DEVICE, "hipMemcpy D2H", e);
    e = hipStreamSynchronize(c->stream);
    if (e != hipSuccess) return fail(c, GS_ERR_DEVICE, "sync", e);
    return GS_OK;
  }
};

static int check_ctx(gs_ctx* c, bool need_crs) {
  if (!c) return GS_ERR_ARG;
  hipError_t e = hipSetDevice(c->device);
  if (e != hipSuccess) return fail(c, GS_ERR_DEVICE, "hipSetDevice", e);
  if (need_crs && !c->have_crs) return fail(c, GS_ERR_NOCRS, "gs_set_crs has not been called");
  return GS_OK;
}

// Matrix<Com>::left_mul for a (k x 1) column: out[i] = sum_j lhs[i][j] col[j].
// Expressed through the engine as `rows` proof-like elements with var tasks on both components.
template <class C, class F>
static int left_mul_impl(gs_ctx* c, int rows, int k, const void* lhs, const void* col, void* out) {
  typedef Fr<C> S;
  size_t com = 2 * (sizeof(Aff<F>) == sizeof(Aff<Fq<C>>) ? Sz<C>::G1 : Sz<C>::G2);
  HostStage st(c);
  void *dl, *dc, *dout;
  RC(st.in(lhs, (size_t)rows * k * sizeof(S), &dl));
  RC(st.in(col, (size_t)k * com, &dc));
  RC(st.out(out, (size_t)rows * com, &dout));
  // pool = canonical lhs via k_prep_verify (m*n scalars)
  PoolMap pm;
  memset(&pm, 0, sizeof pm);
  pm.GC = 0;
  pm.total = rows * k;
  void* pool;
  RC(scratch(c, "lm.pool", (size_t)pm.total * sizeof(S), &pool));
  RC(launch_seg<k_prep_verify<C>>(c, "k_prep_verify", 1, 64, (size_t)1, rows, k, (const S*)dl, (const S*)nullptr,
            (const S*)nullptr, (const S*)nullptr, pm, (S*)pool));
  SidePlan sp;
  int slot = 0;
  for (int i = 0; i < rows; i++) {
    int b[2], e[2];
    for (int a = 0; a < 2; a++) {
      b[a] = slot;
      for (int j = 0; j < k; j++) sp.var.push_back(mkvar(i * k + j, 0, 2 * j + a, slot++));
      e[a] = slot;
    }
    sp.red.push_back(mkred(b[0], e[0], b[1], e[1], 0, i));
  }
  sp.nslots = slot;
  ArrTab arrs;
  memset(&arrs, 0, sizeof arrs);
  arrs.base[0] = (const uint8_t*)dc;
  OutTab outs;
  memset(&outs, 0, sizeof outs);
  outs.base[0] = (uint8_t*)dout;
  RC((run_side<C, F>(c, ".lm", 1, sp, arrs, (const S*)pool, pm.total, (const Aff<F>*)nullptr, outs)));
  return st.back(out, dout, (size_t)rows * com);
}

// Matrix<Fr> product through the prover's own scalar-preparation kernel: with R = lhs^T (inner x rows) and
// Gamma = rhs (inner x cols), Psi = R^T Gamma is lhs * rhs (prove.rs:133; the reference's Mat::right_mul / left_mul on
// Matrix<Fr>, data_structures.rs:824-912).  Shapes with inner * cols >= 1024 take the one-lane-per-output kernel.
template <class C> static int fr_matmul_impl(gs_ctx* c, int rows, int inner, int cols, const void* lhs, const void* rhs, void* out) {
  typedef Fr<C> S;
  std::vector<uint8_t> rt((size_t)inner * rows * sizeof(S));
  const uint8_t* a = (const uint8_t*)lhs;
  for (int k = 0; k < rows; k++)
    for (int i = 0; i < inner; i++) memcpy(&rt[((size_t)i * rows + k) * sizeof(S)], a + ((size_t)k * inner + i) * sizeof(S), sizeof(S));
  HostStage st(c);
  void *dR, *dG, *dout;
  RC(st.in(rt.data(), rt.size(), &dR));
  RC(st.in(rhs, (size_t)inner * cols * sizeof(S), &dG));
  RC(st.out(out, (size_t)rows * cols * sizeof(S), &dout));
  const int m = inner, n = cols, kx = rows, ky = 0;
  PoolMap pm = Impl<C>::prove_pool(m, n, kx, ky);
  void* pool;
  RC(scratch(c, "frmm.pool", (size_t)pm.total * sizeof(S), &pool));
  if (wide_prep(m, n)) {
    int W = m * kx + n * ky + ky * kx + m + n + kx * n + ky * m;
    RC(launch_seg<k_prep_prove_wide_a<C>>(c, "k_prep_prove.a", (size_t)W, 64, (size_t)W, W, m, n, kx, ky, (const S*)dG,
              (const S*)dR, (const S*)dR, (const S*)dR, (const S*)nullptr, (const S*)nullptr, pm, (S*)pool, 0));
  } else {
    RC(launch_seg<k_prep_prove<C>>(c, "k_prep_prove", 1, 64, (size_t)1, m, n, kx, ky, (const S*)dG, (const S*)dR,
              (const S*)dR, (const S*)dR, (const S*)nullptr, (const S*)nullptr, (const S*)nullptr, (const S*)nullptr, pm,
              (S*)pool, 0));
  }
  size_t cnt = (size_t)rows * cols;
  RC(launch(c, "k_fr_to_mont", k_fr_to_mont<C>, cnt, 64, cnt, (const S*)pool + pm.PSI, (S*)dout));
  return st.back(out, dout, cnt * sizeof(S));
}

// ---------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------
// ---- wire format (gs_wire.cuh): arrays of elements, host pointers -------------------------------------
template <class C> struct WireImpl {
  typedef Fq<C> F1;
  typedef Fp2<C> F2;
  template <class F> static int enc_pts(gs_ctx* c, size_t n, int compressed, const void* pts, uint8_t* out) {
    if (n == 0) return GS_OK;
    size_t pb = AFFB(C, F), wb = wire_point_bytes<C, F>(compressed != 0);
    HostStage st(c);
    void *din, *dout;
    RC(st.in(pts, n * pb, &din));
    RC(st.out(out, n * wb, &dout));
    RC(launch(c, "k_wire_enc_pts", k_wire_enc_pts<C, F>, n, 64, n, (const uint8_t*)din, compressed, (uint8_t*)dout));
    return st.back(out, dout, n * wb);
  }
  template <class F>
  static int dec_pts(gs_ctx* c, size_t n, int compressed, int validate, const uint8_t* in, void* pts, uint8_t* ok) {
    if (n == 0) return GS_OK;
    size_t pb = AFFB(C, F), wb = wire_point_bytes<C, F>(compressed != 0);
    HostStage st(c);
    void *din, *dout, *dok;
    RC(st.in(in, n * wb, &din));
    RC(st.out(pts, n * pb, &dout));
    RC(st.out(ok, n, &dok));
    RC(launch(c, "k_wire_dec_pts", k_wire_dec_pts<C, F>, n, 64, n, (const uint8_t*)din, compressed, validate,
              (uint8_t*)dout, (uint8_t*)dok));
    RC(st.back(pts, dout, n * pb));
    return st.back(ok, dok, n);
  }
  // on-curve + r-torsion check of in-memory points (device pointers)
  template <class F> static int validate_pts_dev(gs_ctx* c, size_t n, const void* pts, uint8_t* ok) {
    return launch(c, "k_validate_pts", k_validate_pts<C, F>, n, 64, n, (const uint8_t*)pts, ok);
  }
  // what: 0 Fr, 1 GT
  static int fields(gs_ctx* c, int what, int dir, int validate, size_t n, const void* in, void* out, uint8_t* ok) {
    if (n == 0) return GS_OK;
    size_t per = what == 0 ? 1 : 12, eb = what == 0 ? sizeof(Fr<C>) : 4 * C::N, cnt = n * per;
    HostStage st(c);
    void *din, *dout, *dok = nullptr;
    RC(st.in(in, cnt * eb, &din));
    RC(st.out(out, cnt * eb, &dout));
    std::vector<uint8_t> oks(cnt, 1);
    if (dir == 1) RC(st.out(oks.data(), cnt, &dok));
    if (what == 0)
      RC(launch(c, "k_wire_fr", k_wire_fr<C>, cnt, 64, cnt, dir, (const uint8_t*)din, (uint8_t*)dout, (uint8_t*)dok));
    else
      RC(launch(c, "k_wire_fq", k_wire_fq<C>, cnt, 64, cnt, dir, (const uint8_t*)din, (uint8_t*)dout, (uint8_t*)dok));
    RC(st.back(out, dout, cnt * eb));
    if (dir == 1) {
      RC(st.back(oks.data(), dok, cnt));
      for (size_t i = 0; i < n; i++) {
        uint8_t a = 1;
        for (size_t j = 0; j < per; j++) a &= oks[i * per + j];
        ok[i] = a;
      }
      if (what == 1 && validate) {  // PairingOutput validity: f^r = 1
        void* dok2;
        RC(st.in(ok, n, &dok2));
        RC(launch(c, "k_wire_gt_check", k_wire_gt_check<C>, n, 64, n, (const uint8_t*)dout, (uint8_t*)dok2));
        RC(st.back(ok, dok2, n));
      }
    }
    return GS_OK;
  }
};

extern "C" {

const char* gs_version(void) { return GS_VERSION; }

int gs_sizes(int curve, size_t out[6]) {
  if (curve != 0 && curve != 1) return GS_ERR_ARG;
  size_t fq = sz_fq(curve);
  out[0] = fq;
  out[1] = SZ_FR;
  out[2] = 2 * fq;
  out[3] = 4 * fq;
  out[4] = 12 * fq;
  out[5] = 2 * 4 * fq + 2 * 8 * fq + 2 * fq + 4 * fq + 12 * fq;
  return GS_OK;
}

int gs_ctx_create(int curve, int device, gs_ctx** out) {
  if (!out || (curve != 0 && curve != 1)) return GS_ERR_ARG;
  *out = nullptr;
  int ndev = 0;
  hipError_t e = hipGetDeviceCount(&ndev);
  if (e != hipSuccess || ndev <= 0 || device < 0 || device >= ndev) return GS_ERR_DEVICE;
  if (hipSetDevice(device) != hipSuccess) return GS_ERR_DEVICE;
  gs_ctx* c = new gs_ctx();
  c->curve = curve;
  c->device = device;
  int cus = 0;
  if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && cus > 0)
    c->simd_slots = 4 * (size_t)cus;
  // developer overrides of the planner, documented in include/gs_amd.h next to gs_set_option (which is the API)
  if (const char* e = getenv("GS_COOP_FE")) c->coop_fe = atoi(e);
  if (const char* e = getenv("GS_MILLER_CH")) c->miller_ch = atoi(e);
  if (const char* e = getenv("GS_MILLER_TWIN")) c->miller_twin = atoi(e);
  if (const char* e = getenv("GS_LINE_TABLES")) c->line_tables = atoi(e) != 0;
  if (const char* e = getenv("GS_VAR_TM")) c->var_tm = atoi(e);
  if (const char* e = getenv("GS_VAR_MO")) c->var_mo = atoi(e);
  if (const char* e = getenv("GS_VAR_W")) c->var_w = atoi(e);
  if (const char* e = getenv("GS_RED_K")) c->red_k = atoi(e);
  if (const char* e = getenv("GS_VAR_W2")) c->var_w2 = atoi(e);
  if (const char* e = getenv("GS_ENDO")) c->endo = atoi(e) != 0;
  if (hipEventCreate(&c->ev0) != hipSuccess || hipEventCreate(&c->ev1) != hipSuccess) {
    delete c;
    return GS_ERR_DEVICE;
  }
  for (int i = 0; i < 3 && c->overlap; i++)
    if (hipStreamCreateWithFlags(&c->side[i], hipStreamNonBlocking) != hipSuccess) c->overlap = false;
  for (int i = 0; i < 6 && c->overlap; i++)
    if (hipEventCreateWithFlags(&c->sev[i], hipEventDisableTiming) != hipSuccess) c->overlap = false;
  if (!c->overlap) c->side[0] = nullptr;
  if (const char* e = getenv("GS_OVERLAP")) c->overlap = c->overlap && atoi(e) != 0;
  {
    std::lock_guard<std::mutex> lk(g_ctx_mu);
    g_ctx_all.push_back(c);
  }
  *out = c;
  return GS_OK;
}

void gs_ctx_destroy(gs_ctx* c) {
  if (!c) return;
  {
    std::lock_guard<std::mutex> lk(g_ctx_mu);
    g_ctx_all.erase(std::remove(g_ctx_all.begin(), g_ctx_all.end(), c), g_ctx_all.end());
  }
  hipSetDevice(c->device);
  drain_ctx(c);
  for (void* p : c->deferred_free) hipFree(p);
  if (c->stamp) hipFree(c->stamp);
  for (auto& kv : c->scratch)
    if (kv.second.p) hipFree(kv.second.p);
  for (hipStream_t st : c->side)
    if (st) hipStreamSynchronize(st);
  for (auto& kv : c->plans)
    if (kv.second.dev.p) hipFree(kv.second.dev.p);
  delete c->pool;
  if (c->pin) hipHostFree(c->pin);
  if (c->copy_stream) hipStreamDestroy(c->copy_stream);
  for (hipStream_t st : {c->copy_stream2, c->copy_out_stream, c->copy_out_stream2})
    if (st) hipStreamDestroy(st);
  for (hipEvent_t ev : c->pev)
    if (ev) hipEventDestroy(ev);
  for (hipEvent_t ev : c->pev2)
    if (ev) hipEventDestroy(ev);
  if (c->pev_ready) hipEventDestroy(c->pev_ready);
  c->tabs.reset();  // the CRS tables go with their last context
  for (hipStream_t st : c->side)
    if (st) hipStreamDestroy(st);
  for (hipEvent_t ev : c->sev)
    if (ev) hipEventDestroy(ev);
  if (c->ev0) hipEventDestroy(c->ev0);
  if (c->ev1) hipEventDestroy(c->ev1);
  delete c;
}

int gs_set_stream(gs_ctx* c, void* s) {
  if (!c) return GS_ERR_ARG;
  c->stream = (hipStream_t)s;
  return GS_OK;
}
int gs_set_option(gs_ctx* c, const char* key, int value) {
  if (!c || !key) return GS_ERR_ARG;
  std::string k(key);
  if (k == "miller_twin") {
    if (value < -1 || value > 3)
      return fail(c, GS_ERR_ARG, "miller_twin: -1 (planned), 0 single, 1 twin, 2 lane pair (LDS), 3 lane pair (DPP)");
    c->miller_twin = value;
  } else if (k == "var_tab") {
    if (value < -1 || value > 1) return fail(c, GS_ERR_ARG, "var_tab: -1 (planned), 0 Straus lanes, 1 shared per-base window tables");
    c->var_tab = value;
  } else if (k == "var_w2") {
    if (value < -1 || value > 1) return fail(c, GS_ERR_ARG, "var_w2: -1 (planned), 0 one wave per SIMD, 1 two (G1 Straus lanes)");
    c->var_w2 = value;
  } else if (k == "endo") {
    c->endo = value != 0;
  } else if (k == "mixed_merge") {
    if (value < -1 || value > 1) return fail(c, GS_ERR_ARG, "mixed_merge: -1 (planned), 0 parts one after the other, 1 merged launches");
    c->mixed_merge = value;
  } else if (k == "miller_ch") {
    if (value < 0 || value > MILLER_CH) return fail(c, GS_ERR_ARG, "miller_ch: 0 (planned) .. capacity of a Miller lane");
    c->miller_ch = value;
  } else if (k == "var_tm") {
    if (value < 0 || value > 8) return fail(c, GS_ERR_ARG, "var_tm: 0 (planned) .. 8");
    c->var_tm = value;
  } else if (k == "var_mo") {
    if (value != 0 && value != 1 && value != 2 && value != 4) return fail(c, GS_ERR_ARG, "var_mo: 0 (planned), 1, 2, 4");
    c->var_mo = value;
  } else if (k == "var_w") {
    if (value != 0 && value != 4 && value != 5) return fail(c, GS_ERR_ARG, "var_w: 0 (planned), 4, 5");
    c->var_w = value;
  } else if (k == "var_ws_lanes") {
    if (value < 0 || (value > 0 && value < 64) || value % 64) return fail(c, GS_ERR_ARG, "var_ws_lanes: 0 or a multiple of 64");
    c->var_ws_lanes = value;
  } else if (k == "red_k") {
    if (value != 0 && value != 1 && value != 2 && value != 4 && value != 8)
      return fail(c, GS_ERR_ARG, "red_k: 0 (planned), 1, 2, 4, 8");
    c->red_k = value;
  } else if (k == "coop_fe") {
    if (value < 0 || value > 2) return fail(c, GS_ERR_ARG, "coop_fe: 0 never, 1 planned, 2 always");
    c->coop_fe = value;
  } else if (k == "line_tables") {
    c->line_tables = value != 0;
  } else if (k == "overlap") {
    c->overlap = value != 0 && c->side[0] != nullptr;
  } else {
    return fail(c, GS_ERR_ARG, "unknown option");
  }
  return GS_OK;
}
int gs_sync(gs_ctx* c) {
  RC(check_ctx(c, false));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return GS_OK;
}
const char* gs_last_error(gs_ctx* c) { return c ? c->err.c_str() : "null context"; }

// Page-lock a caller's buffer (a Vec the Rust side reuses from call to call): the host-pointer entry points then move
// it by DMA directly instead of through the staging copy.  The runtime counts registrations of a range silently; the
// library keeps its own list so that a second registration, or the release of an unknown pointer, is an error.
int gs_host_register(gs_ctx* c, void* ptr, size_t bytes) {
  RC(check_ctx(c, false));
  if (!ptr || !bytes) return GS_ERR_ARG;
  std::lock_guard<std::mutex> lk(g_reg_mu);
  if (g_reg.count((const uint8_t*)ptr)) return fail(c, GS_ERR_ARG, "gs_host_register: already registered");
  hipError_t e = hipHostRegister(ptr, bytes, hipHostRegisterPortable);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    return fail(c, e == hipErrorOutOfMemory ? GS_ERR_ALLOC : GS_ERR_ARG, "hipHostRegister", e);
  }
  g_reg[(const uint8_t*)ptr] = bytes;
  return GS_OK;
}
int gs_host_unregister(gs_ctx* c, void* ptr) {
  RC(check_ctx(c, false));
  if (!ptr) return GS_ERR_ARG;
  std::lock_guard<std::mutex> lk(g_reg_mu);
  if (!g_reg.count((const uint8_t*)ptr)) return fail(c, GS_ERR_ARG, "gs_host_unregister: not registered here");
  // Nothing may still be moving the buffer: the registry is process-wide, so EVERY live context is drained -- its
  // compute stream and its copy queues (a D2H is only gated on the compute stream, not finished with it), the shards
  // of a gs_multi context included (they are contexts of their own on their own devices).
  {
    std::lock_guard<std::mutex> lk2(g_ctx_mu);
    for (gs_ctx* o : g_ctx_all) {
      hipSetDevice(o->device);
      drain_ctx(o);
    }
    hipSetDevice(c->device);
  }
  hipError_t e = hipHostUnregister(ptr);
  g_reg.erase((const uint8_t*)ptr);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    return fail(c, GS_ERR_ARG, "hipHostUnregister", e);
  }
  return GS_OK;
}

// Buffers for a caller that wants the fast path without thinking about it: a mapping of its own
// (anonymous mmap: page-aligned, never part of the allocator's heap, so no stale runtime pin can alias it), every page
// touched up front (MAP_POPULATE: no first-touch faults inside a call -- 27-35 ms at 2^16 for result arrays made per
// call), page-locked and entered in the registry: arrays inside it go by DMA straight from / to the caller's memory.
static std::mutex g_alloc_mu;
static std::map<void*, size_t> g_alloc;  // gs_host_alloc: start -> mapped bytes
int gs_host_alloc(gs_ctx* c, size_t bytes, void** out) {
  RC(check_ctx(c, false));
  if (!out || !bytes) return GS_ERR_ARG;
  *out = nullptr;
  const size_t page = 4096, len = (bytes + page - 1) / page * page;
  void* p = mmap(nullptr, len, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_POPULATE, -1, 0);
  if (p == MAP_FAILED) return fail(c, GS_ERR_ALLOC, "gs_host_alloc: mmap");
  int rc = gs_host_register(c, p, len);
  if (rc != GS_OK) {
    munmap(p, len);
    return rc;
  }
  {
    std::lock_guard<std::mutex> lk(g_alloc_mu);
    g_alloc[p] = len;
  }
  *out = p;
  return GS_OK;
}
int gs_host_free(gs_ctx* c, void* ptr) {
  RC(check_ctx(c, false));
  if (!ptr) return GS_ERR_ARG;
  size_t len = 0;
  {
    std::lock_guard<std::mutex> lk(g_alloc_mu);
    auto it = g_alloc.find(ptr);
    if (it == g_alloc.end()) return fail(c, GS_ERR_ARG, "gs_host_free: not a gs_host_alloc buffer");
    len = it->second;
    g_alloc.erase(it);
  }
  int rc = gs_host_unregister(c, ptr);  // drains every context first
  munmap(ptr, len);
  return rc;
}

int gs_set_crs(gs_ctx* c, const void* crs) {
  RC(check_ctx(c, false));
  if (!crs) return GS_ERR_ARG;
  return DISPATCH(c, set_crs(c, crs));
}

// ---- CRS generation (generator.rs:81-118) --------------------------------------------
int gs_crs_generate(gs_ctx* c, const void* p1, const void* p2, const void* sc, void* out) {
  RC(check_ctx(c, false));
  if (!p1 || !p2 || !sc || !out) return fail(c, GS_ERR_ARG, "null pointer");
  size_t fq = sz_fq(c->curve), g1 = 2 * fq, g2 = 4 * fq;
  const uint8_t* s = (const uint8_t*)sc;  // a1, a2, t1, t2
  uint8_t* o = (uint8_t*)out;
  std::vector<uint8_t> q1(g1), u1(g1), v1(g1), q2(g2), u2(g2), v2(g2);
  // q = a*p, u = t*p, v = t*q   (prepare_real_binding_key: v = q.mul(t) - 0)
  RC(gs_g1_mul_batch(c, 1, p1, 1, s + 0 * SZ_FR, q1.data()));
  RC(gs_g1_mul_batch(c, 1, p1, 1, s + 2 * SZ_FR, u1.data()));
  RC(gs_g1_mul_batch(c, 1, q1.data(), 1, s + 2 * SZ_FR, v1.data()));
  RC(gs_g2_mul_batch(c, 1, p2, 1, s + 1 * SZ_FR, q2.data()));
  RC(gs_g2_mul_batch(c, 1, p2, 1, s + 3 * SZ_FR, u2.data()));
  RC(gs_g2_mul_batch(c, 1, q2.data(), 1, s + 3 * SZ_FR, v2.data()));
  memcpy(o, p1, g1); o += g1;
  memcpy(o, q1.data(), g1); o += g1;
  memcpy(o, u1.data(), g1); o += g1;
  memcpy(o, v1.data(), g1); o += g1;
  memcpy(o, p2, g2); o += g2;
  memcpy(o, q2.data(), g2); o += g2;
  memcpy(o, u2.data(), g2); o += g2;
  memcpy(o, v2.data(), g2); o += g2;
  memcpy(o, p1, g1); o += g1;
  memcpy(o, p2, g2); o += g2;
  return gs_multi_pairing_batch(c, 1, 1, p1, p2, o);
}

// The simulation ("hiding") key of generator.rs:65-77: identical except v = t*q - generator, so that commitments are
// perfectly hiding.  t*q - p is one left_mul with the row (t, -1) over the column ((q, q), (p, p)).
int gs_crs_generate_hiding(gs_ctx* c, const void* p1, const void* p2, const void* sc, void* out) {
  RC(gs_crs_generate(c, p1, p2, sc, out));
  size_t fq = sz_fq(c->curve), g1 = 2 * fq, g2 = 4 * fq;
  const uint8_t* s = (const uint8_t*)sc;
  uint8_t* o = (uint8_t*)out;
  // Montgomery form of -1: r - (2^256 mod r)
  uint32_t m1[8];
  {
    uint64_t br = 0;
    for (int i = 0; i < 8; i++) {
      uint32_t rw = c->curve == 0 ? Bls12_381::R_WORDS[i] : Bn254::R_WORDS[i];
      uint32_t ow = c->curve == 0 ? Bls12_381::Q_ONE[i] : Bn254::Q_ONE[i];
      uint64_t x = (uint64_t)rw - ow - br;
      m1[i] = (uint32_t)x;
      br = (x >> 63) & 1;
    }
  }
  std::vector<uint8_t> lhs(2 * SZ_FR), col, res;
  memcpy(lhs.data() + SZ_FR, m1, SZ_FR);
  // G1: out layout u0 = (p1, q1), u1 = (t1 p1, v1)
  memcpy(lhs.data(), s + 2 * SZ_FR, SZ_FR);
  col.assign(4 * g1, 0);
  res.assign(2 * g1, 0);
  memcpy(col.data(), o + g1, g1);           // (q1, q1)
  memcpy(col.data() + g1, o + g1, g1);
  memcpy(col.data() + 2 * g1, p1, g1);      // (p1, p1)
  memcpy(col.data() + 3 * g1, p1, g1);
  RC(gs_mat_left_mul_com1(c, 1, 2, lhs.data(), col.data(), res.data()));
  memcpy(o + 3 * g1, res.data(), g1);
  // G2: v0 = (p2, q2), v1 = (t2 p2, v2) after the 4 G1 points
  uint8_t* o2 = o + 4 * g1;
  memcpy(lhs.data(), s + 3 * SZ_FR, SZ_FR);
  col.assign(4 * g2, 0);
  res.assign(2 * g2, 0);
  memcpy(col.data(), o2 + g2, g2);
  memcpy(col.data() + g2, o2 + g2, g2);
  memcpy(col.data() + 2 * g2, p2, g2);
  memcpy(col.data() + 3 * g2, p2, g2);
  RC(gs_mat_left_mul_com2(c, 1, 2, lhs.data(), col.data(), res.data()));
  memcpy(o2 + 3 * g2, res.data(), g2);
  return GS_OK;
}

// ---- commit ------------------------------------------------------------------
#define COMMIT_DEV(NAME, FT, GROUP, TAB, TAG)                                                              \
  int NAME(gs_ctx* c, size_t n, const void* v, const void* r, void* out) {                                 \
    RC(check_ctx(c, true));                                                                                \
    if (n == 0) return GS_OK;                                                                              \
    if (!v || !r || !out) return GS_ERR_ARG;                                                               \
    if (c->curve == 0)                                                                                     \
      return Impl<Bls12_381>::commit<FT<Bls12_381>>(c, n, GROUP, v, r, out, (const Aff<FT<Bls12_381>>*)c->tabs->TAB.p, TAG); \
    return Impl<Bn254>::commit<FT<Bn254>>(c, n, GROUP, v, r, out, (const Aff<FT<Bn254>>*)c->tabs->TAB.p, TAG);   \
  }
COMMIT_DEV(gs_commit_g1_dev, Fq, true, tab16_g1, ".cg1")
COMMIT_DEV(gs_commit_g2_dev, Fp2, true, tab16_g2, ".cg2")
COMMIT_DEV(gs_commit_fr_b1_dev, Fq, false, tab16_g1, ".cg1")
COMMIT_DEV(gs_commit_fr_b2_dev, Fp2, false, tab16_g2, ".cg2")

#define COMMIT_HOST(NAME, DEVNAME, VSZ, KC, OSZ)                           \
  int NAME(gs_ctx* c, size_t n, const void* v, const void* r, void* out) { \
    RC(check_ctx(c, true));                                                \
    if (n == 0) return GS_OK;                                              \
    if (!v || !r || !out) return GS_ERR_ARG;                               \
    size_t fq = sz_fq(c->curve);                                           \
    HostStage st(c);                                                       \
    void *dv, *dr, *dout;                                                  \
    RC(st.in(v, n * (VSZ), &dv));                                          \
    RC(st.in(r, n * (KC)*SZ_FR, &dr));                                     \
    RC(st.out(out, n * (OSZ), &dout));                                     \
    RC(DEVNAME(c, n, dv, dr, dout));                                       \
    return st.back(out, dout, n * (OSZ));                                  \
  }
COMMIT_HOST(gs_commit_g1, gs_commit_g1_dev, 2 * fq, 2, 4 * fq)
COMMIT_HOST(gs_commit_g2, gs_commit_g2_dev, 4 * fq, 2, 8 * fq)
COMMIT_HOST(gs_commit_fr_b1, gs_commit_fr_b1_dev, SZ_FR, 1, 4 * fq)
COMMIT_HOST(gs_commit_fr_b2, gs_commit_fr_b2_dev, SZ_FR, 1, 8 * fq)

// ---- prove -------------------------------------------------------------------
static int check_shape(gs_ctx* c, int ty, int m, int n) {
  if (ty < 0 || ty > 3) return fail(c, GS_ERR_ARG, "bad equation type");
  // the reference indexes rand[0] / gamma[0]: empty variable lists panic (prove.rs:106-113)
  if (m < 1 || n < 1) return fail(c, GS_ERR_SHAPE, "m and n must be >= 1 (reference asserts, prove.rs:106-113)");
  if (m > 4096 || n > 4096 || (long)m * n > (1L << 22)) return fail(c, GS_ERR_SHAPE, "shape too large for task tables");
  return GS_OK;
}

int gs_prove_batch_dev(gs_ctx* c, int ty, size_t N, int m, int n, const void* X, const void* Y, const void* A,
                       const void* B, const void* G, const void* R, const void* S, const void* T, void* xcoms,
                       void* ycoms, void* pi, void* theta) {
  RC(check_ctx(c, true));
  RC(check_shape(c, ty, m, n));
  if (N == 0) return GS_OK;
  if (!X || !Y || !A || !B || !G || !R || !S || !T || !pi || !theta) return fail(c, GS_ERR_ARG, "null pointer");
  return DISPATCH(c, prove(c, ty, N, m, n, X, Y, A, B, G, R, S, T, xcoms, ycoms, pi, theta));
}

// host-pointer prove in three phases so that a mixed call can interleave several sub-batches: stage (the workers start
// copying), run (kernels enqueued behind per-array upload events), finish (outputs back, stream drained)
struct ProveArgs {
  int ty;
  size_t N;
  int m, n;
  const void *X, *Y, *A, *B, *G, *R, *S, *T;
  void *xcoms, *ycoms, *pi, *theta;
  bool shared;  // a Statement's part: ONE copy of X, Y, R, S and of the commitments for all N equations
};
static int prove_host_stage(gs_ctx* c, const ProveArgs& a, HostPipe& hp, size_t base = 0, bool start = true) {
  size_t fq = sz_fq(c->curve), N = a.N;
  bool xg = x_is_group(a.ty), yg = y_is_group(a.ty);
  int kx = xg ? 2 : 1, ky = yg ? 2 : 1, m = a.m, n = a.n;
  size_t sx = xg ? 2 * fq : SZ_FR, sy = yg ? 4 * fq : SZ_FR;
  const size_t V = a.shared ? 1 : N;
  hp.in(PI_X, a.X, V * m * sx);
  hp.in(PI_Y, a.Y, V * n * sy);
  hp.in(PI_A, a.A, N * n * sx);
  hp.in(PI_B, a.B, N * m * sy);
  hp.in(PI_G, a.G, N * m * n * SZ_FR);
  hp.in(PI_R, a.R, V * m * kx * SZ_FR);
  hp.in(PI_S, a.S, V * n * ky * SZ_FR);
  hp.in(PI_T, a.T, N * ky * kx * SZ_FR);
  hp.out(PO_XC, a.xcoms, V * m * 4 * fq);
  hp.out(PO_YC, a.ycoms, V * n * 8 * fq);
  hp.out(PO_PI, a.pi, N * kx * 8 * fq);
  hp.out(PO_TH, a.theta, N * ky * 4 * fq);
  // staging order = the order prove() asks for them (scalars, then the G2 side's arguments, then the G1 side's)
  static const int order[] = {PI_G, PI_R, PI_S, PI_T, PI_Y, PI_B, PI_X, PI_A};  // (Y before B: the fixed-base kernel reads Y)
  return start ? hp.begin(order, 8, base, base == 0) : (int)GS_OK;
}
static int prove_host_run(gs_ctx* c, const ProveArgs& a, HostPipe& hp) {
  return (a.shared ? gs_prove_statement_dev : gs_prove_batch_dev)(c, a.ty, a.N, a.m, a.n, hp.dev(PI_X), hp.dev(PI_Y), hp.dev(PI_A), hp.dev(PI_B), hp.dev(PI_G),
                            hp.dev(PI_R), hp.dev(PI_S), hp.dev(PI_T), hp.dev(PO_XC), hp.dev(PO_YC), hp.dev(PO_PI),
                            hp.dev(PO_TH));
}
int gs_prove_batch(gs_ctx* c, int ty, size_t N, int m, int n, const void* X, const void* Y, const void* A,
                   const void* B, const void* G, const void* R, const void* S, const void* T, void* xcoms,
                   void* ycoms, void* pi, void* theta) {
  RC(check_ctx(c, true));
  RC(check_shape(c, ty, m, n));
  if (N == 0) return GS_OK;
  if (!X || !Y || !A || !B || !G || !R || !S || !T || !pi || !theta) return fail(c, GS_ERR_ARG, "null pointer");
  ProveArgs a{ty, N, m, n, X, Y, A, B, G, R, S, T, xcoms, ycoms, pi, theta, false};
  HostPipe hp(c);
  RC(prove_host_stage(c, a, hp));
  RC(prove_host_run(c, a, hp));
  return hp.finish();
}

// ---- verify ------------------------------------------------------------------
int gs_verify_batch_dev(gs_ctx* c, int ty, size_t N, int m, int n, const void* A, const void* B, const void* G,
                        const void* target, const void* xcoms, const void* ycoms, const void* pi, const void* theta,
                        uint8_t* ok) {
  RC(check_ctx(c, true));
  RC(check_shape(c, ty, m, n));
  if (N == 0) return GS_OK;
  if (!A || !B || !G || !target || !xcoms || !ycoms || !pi || !theta || !ok) return fail(c, GS_ERR_ARG, "null pointer");
  return DISPATCH(c, verify(c, ty, N, m, n, A, B, G, target, xcoms, ycoms, pi, theta, ok));
}

struct VerifyArgs {
  int ty;
  size_t N;
  int m, n;
  const void *A, *B, *G, *target, *xcoms, *ycoms, *pi, *theta;
  uint8_t* ok;
  bool shared;
};
static int verify_host_stage(gs_ctx* c, const VerifyArgs& a, HostPipe& hp, size_t base = 0, bool start = true) {
  size_t fq = sz_fq(c->curve), N = a.N;
  int ty = a.ty, m = a.m, n = a.n;
  bool xg = x_is_group(ty), yg = y_is_group(ty);
  int kx = xg ? 2 : 1, ky = yg ? 2 : 1;
  size_t sx = xg ? 2 * fq : SZ_FR, sy = yg ? 4 * fq : SZ_FR;
  size_t st_ = ty == GS_PPE ? 12 * fq : ty == GS_MSMEG1 ? 2 * fq : ty == GS_MSMEG2 ? 4 * fq : SZ_FR;
  hp.in(VI_A, a.A, N * n * sx);
  hp.in(VI_B, a.B, N * m * sy);
  hp.in(VI_G, a.G, N * m * n * SZ_FR);
  hp.in(VI_TG, a.target, N * st_);
  hp.in(VI_XC, a.xcoms, (a.shared ? 1 : N) * m * 4 * fq);
  hp.in(VI_YC, a.ycoms, (a.shared ? 1 : N) * n * 8 * fq);
  hp.in(VI_PI, a.pi, N * kx * 8 * fq);
  hp.in(VI_TH, a.theta, N * ky * 4 * fq);
  hp.out(VO_OK, a.ok, N);
  // Gamma and the G1-side arguments first (the verifier's Gamma-MSM runs under the upload of the rest)
  static const int order_g[] = {VI_G, VI_A, VI_XC, VI_TG, VI_B, VI_YC, VI_PI, VI_TH};
  static const int order_s[] = {VI_G, VI_A, VI_B, VI_TG, VI_XC, VI_YC, VI_PI, VI_TH};
  return start ? hp.begin(ty == GS_PPE || ty == GS_MSMEG2 ? order_g : order_s, 8, base, base == 0) : (int)GS_OK;
}
static int verify_host_run(gs_ctx* c, const VerifyArgs& a, HostPipe& hp) {
  return (a.shared ? gs_verify_statement_dev : gs_verify_batch_dev)(c, a.ty, a.N, a.m, a.n, hp.dev(VI_A), hp.dev(VI_B), hp.dev(VI_G), hp.dev(VI_TG),
                             hp.dev(VI_XC), hp.dev(VI_YC), hp.dev(VI_PI), hp.dev(VI_TH), (uint8_t*)hp.dev(VO_OK));
}
int gs_verify_batch(gs_ctx* c, int ty, size_t N, int m, int n, const void* A, const void* B, const void* G,
                    const void* target, const void* xcoms, const void* ycoms, const void* pi, const void* theta,
                    uint8_t* ok) {
  RC(check_ctx(c, true));
  RC(check_shape(c, ty, m, n));
  if (N == 0) return GS_OK;
  if (!A || !B || !G || !target || !xcoms || !ycoms || !pi || !theta || !ok) return fail(c, GS_ERR_ARG, "null pointer");
  VerifyArgs a{ty, N, m, n, A, B, G, target, xcoms, ycoms, pi, theta, ok, false};
  HostPipe hp(c);
  RC(verify_host_stage(c, a, hp));
  RC(verify_host_run(c, a, hp));
  return hp.finish();
}

// ---- mixed batches: several sub-batches (any types and shapes) in ONE call ----------------------------------------
// configs[2] of the baseline is a batch of PPE, MSMEG1 and MSMEG2 equations; the reference's Statement is a list of
// equations of any type (statement.rs:24-28).  The parts run one after the other on the context's stream.
// Measured and NOT kept (profiles/r3/mixed_streams.txt): one child context per part (own stream, own scratch) so that
// the parts' kernels are in flight together.  These kernels hold one 512-register wave per SIMD and a private segment
// each; dispatches from different queues did not share the chip at all (2^12 mixed: 71.1 ms on three streams, 71.3 ms
// on one), and the extra queues' scratch reservations slowed LATER large batches on the parent fourfold (2^16 PPE
// 320 -> 1400 ms).  What does fill the chip for a mixed batch of a few thousand equations is merging the lanes of
// all parts into single launches (segmented launches, section 4.3 of DESIGN.md).
// Record every part's launches (nothing is enqueued yet), then replay them merged.  A single part simply runs.  The
// kernel profile (gs_prof_enable) times the MERGED launches themselves (go_seg: HIP events around each k_seg launch, under
// the name of its first segment), so that a mixed step's per-kernel times add up to the step that is timed.
// Merged launches pay off while the parts' own launches leave SIMDs idle; from ~2^15 equations on every part fills the
// chip by itself and separate launches with per-part lane shapes are as fast (measured: profiles/r3/mixed_merge.txt).
static bool merge_parts(const gs_ctx* c, size_t total_n) {
  if (c->mixed_merge >= 0) return c->mixed_merge != 0;
  // <= 2^14 equations on a 256-CU device (profiles/r3/mixed_merge.txt: merged 103 against 112 ms in sequence at 2^14,
  // 193 against 182 ms at 2^15, where every part fills the chip with its own lane shapes)
  return total_n <= 16 * c->simd_slots;
}
// buffers parked by ensure() during a recording: the replayed launches may read them, so they go after a stream sync
static int release_deferred(gs_ctx* c, int rc) {
  if (c->deferred_free.empty()) return rc;
  hipStreamSynchronize(c->stream);
  for (void* p : c->deferred_free) hipFree(p);
  c->deferred_free.clear();
  return rc;
}
extern "C++" {
template <class FN> static int mixed_run(gs_ctx* c, int nparts, size_t total_n, FN part_fn) {
  if (nparts <= 1 || c->rec || !merge_parts(c, total_n)) {
    for (int i = 0; i < nparts; i++) RC(part_fn(i));
    return GS_OK;
  }
  Recorder R;
  R.parts.resize(nparts);
  c->rec = &R;
  c->fill_n = total_n;
  int rc = GS_OK;
  for (int i = 0; i < nparts && rc == GS_OK; i++) {
    R.cur = i;
    c->scratch_tag = i + 1;
    rc = part_fn(i);
  }
  c->rec = nullptr;
  c->fill_n = 0;
  c->scratch_tag = 0;
  if (rc == GS_OK) rc = replay(c, R);  // (a failed part: nothing was launched)
  return release_deferred(c, rc);
}
}  // extern "C++"
int gs_prove_mixed_dev(gs_ctx* c, int nparts, const gs_prove_part* p) {
  RC(check_ctx(c, true));
  if (nparts < 0 || nparts > GS_MIXED_MAX || (nparts && !p)) return fail(c, GS_ERR_ARG, "gs_prove_mixed: 0..8 parts");
  RangeGuard rg("gs.prove_mixed");
  size_t tot = 0;
  for (int i = 0; i < nparts; i++) tot += p[i].N;
  return mixed_run(c, nparts, tot, [&](int i) {
    return (p[i].shared_vars ? gs_prove_statement_dev : gs_prove_batch_dev)(
        c, p[i].equ_type, p[i].N, p[i].m, p[i].n, p[i].X, p[i].Y, p[i].A, p[i].B, p[i].Gamma, p[i].R, p[i].S, p[i].T,
        p[i].xcoms, p[i].ycoms, p[i].pi, p[i].theta);
  });
}
int gs_verify_mixed_dev(gs_ctx* c, int nparts, const gs_verify_part* p) {
  RC(check_ctx(c, true));
  if (nparts < 0 || nparts > GS_MIXED_MAX || (nparts && !p)) return fail(c, GS_ERR_ARG, "gs_verify_mixed: 0..8 parts");
  RangeGuard rg("gs.verify_mixed");
  size_t tot = 0;
  for (int i = 0; i < nparts; i++) tot += p[i].N;
  return mixed_run(c, nparts, tot, [&](int i) {
    return (p[i].shared_vars ? gs_verify_statement_dev : gs_verify_batch_dev)(
        c, p[i].equ_type, p[i].N, p[i].m, p[i].n, p[i].A, p[i].B, p[i].Gamma, p[i].target, p[i].xcoms, p[i].ycoms,
        p[i].pi, p[i].theta, p[i].ok);
  });
}
// host pointers: every part is staged into its own region of the pinned buffer (the memcpy workers start at once), the
// parts' launches are recorded behind their upload events and replayed merged, then the outputs come back
extern "C++" {
template <class ARGS, class STAGE, class RUN>
static int mixed_host(gs_ctx* c, std::vector<ARGS>& a, size_t total_n, STAGE stage, RUN run) {
  const int np = (int)a.size();
  if (np <= 1 || !merge_parts(c, total_n)) {
    for (int i = 0; i < np; i++) {
      HostPipe h(c);
      RC(stage(c, a[i], h, (size_t)0, true));
      RC(run(c, a[i], h));
      RC(h.finish());
    }
    return GS_OK;
  }
  std::vector<std::unique_ptr<HostPipe>> hp;
  // 1. how much pinned memory the call needs: declare the arrays on throw-away pipes
  size_t need_pin = 0;
  std::vector<size_t> base(np, 0);
  for (int i = 0; i < np; i++) {
    hp.emplace_back(new HostPipe(c));
    base[i] = need_pin;
    need_pin += stage(c, a[i], *hp[i], (size_t)0, false) == GS_OK ? hp[i]->layout() : 0;
  }
  hp.clear();
  c->pipe = nullptr;
  RC(HostPipe::pin_reserve(c, need_pin));
  // 2. stage for real, every part at its base
  int rc = GS_OK;
  for (int i = 0; i < np && rc == GS_OK; i++) {
    hp.emplace_back(new HostPipe(c));
    c->scratch_tag = i + 1;
    rc = stage(c, a[i], *hp[i], base[i], true);
  }
  // 3. record the parts' launches (their need() calls enqueue the uploads and make the stream wait), replay merged
  Recorder R;
  R.parts.resize(np);
  c->rec = &R;
  c->fill_n = total_n;
  for (int i = 0; i < np && rc == GS_OK; i++) {
    R.cur = i;
    c->scratch_tag = i + 1;
    c->pipe = hp[i].get();
    rc = run(c, a[i], *hp[i]);
  }
  c->rec = nullptr;
  c->fill_n = 0;
  c->scratch_tag = 0;
  c->pipe = nullptr;
  if (rc == GS_OK) rc = replay(c, R);
  // 4. outputs
  for (int i = 0; i < np; i++) {
    if (rc == GS_OK) {
      c->scratch_tag = i + 1;
      rc = hp[i]->finish();
    }
  }
  c->scratch_tag = 0;
  if (rc != GS_OK) drain_ctx(c);  // uploads / downloads of the parts that did start (not only c->stream)
  return release_deferred(c, rc);
}
}  // extern "C++"
int gs_prove_mixed(gs_ctx* c, int nparts, const gs_prove_part* p) {
  RC(check_ctx(c, true));
  if (nparts < 0 || nparts > GS_MIXED_MAX || (nparts && !p)) return fail(c, GS_ERR_ARG, "gs_prove_mixed: 0..8 parts");
  std::vector<ProveArgs> a;
  size_t tot = 0;
  for (int i = 0; i < nparts; i++) {
    RC(check_shape(c, p[i].equ_type, p[i].m, p[i].n));
    if (p[i].N == 0) continue;
    if (!p[i].X || !p[i].Y || !p[i].A || !p[i].B || !p[i].Gamma || !p[i].R || !p[i].S || !p[i].T || !p[i].pi || !p[i].theta)
      return fail(c, GS_ERR_ARG, "null pointer");
    a.push_back(ProveArgs{p[i].equ_type, p[i].N, p[i].m, p[i].n, p[i].X, p[i].Y, p[i].A, p[i].B, p[i].Gamma, p[i].R,
                          p[i].S, p[i].T, p[i].xcoms, p[i].ycoms, p[i].pi, p[i].theta, p[i].shared_vars != 0});
    tot += p[i].N;
  }
  RangeGuard rg("gs.prove_mixed");
  return mixed_host(c, a, tot, prove_host_stage, prove_host_run);
}
int gs_verify_mixed(gs_ctx* c, int nparts, const gs_verify_part* p) {
  RC(check_ctx(c, true));
  if (nparts < 0 || nparts > GS_MIXED_MAX || (nparts && !p)) return fail(c, GS_ERR_ARG, "gs_verify_mixed: 0..8 parts");
  std::vector<VerifyArgs> a;
  size_t tot = 0;
  for (int i = 0; i < nparts; i++) {
    RC(check_shape(c, p[i].equ_type, p[i].m, p[i].n));
    if (p[i].N == 0) continue;
    if (!p[i].A || !p[i].B || !p[i].Gamma || !p[i].target || !p[i].xcoms || !p[i].ycoms || !p[i].pi || !p[i].theta || !p[i].ok)
      return fail(c, GS_ERR_ARG, "null pointer");
    a.push_back(VerifyArgs{p[i].equ_type, p[i].N, p[i].m, p[i].n, p[i].A, p[i].B, p[i].Gamma, p[i].target, p[i].xcoms,
                           p[i].ycoms, p[i].pi, p[i].theta, p[i].ok, p[i].shared_vars != 0});
    tot += p[i].N;
  }
  RangeGuard rg("gs.verify_mixed");
  return mixed_host(c, a, tot, verify_host_stage, verify_host_run);
}

// ---- Statement: E equations of one type over the SAME committed variables (statement.rs:24-28,109) -------------
int gs_prove_statement_dev(gs_ctx* c, int ty, size_t E, int m, int n, const void* X, const void* Y, const void* A,
                           const void* B, const void* G, const void* R, const void* S, const void* T, void* xcoms,
                           void* ycoms, void* pi, void* theta) {
  RC(check_ctx(c, true));
  RC(check_shape(c, ty, m, n));
  if (E == 0) return GS_OK;
  if (!X || !Y || !A || !B || !G || !R || !S || !T || !pi || !theta) return fail(c, GS_ERR_ARG, "null pointer");
  bool xg = x_is_group(ty), yg = y_is_group(ty);
  // the commitments, once (commit.rs:78-100,125-156,178-200,225-256)
  RC(need(c, BIT(PI_X) | BIT(PI_Y) | BIT(PI_R) | BIT(PI_S)));
  if (xcoms) RC((xg ? gs_commit_g1_dev : gs_commit_fr_b1_dev)(c, (size_t)m, X, R, xcoms));
  if (ycoms) RC((yg ? gs_commit_g2_dev : gs_commit_fr_b2_dev)(c, (size_t)n, Y, S, ycoms));
  return DISPATCH(c, prove(c, ty, E, m, n, X, Y, A, B, G, R, S, T, nullptr, nullptr, pi, theta, true));
}
int gs_verify_statement_dev(gs_ctx* c, int ty, size_t E, int m, int n, const void* A, const void* B, const void* G,
                            const void* target, const void* xcoms, const void* ycoms, const void* pi,
                            const void* theta, uint8_t* ok) {
  RC(check_ctx(c, true));
  RC(check_shape(c, ty, m, n));
  if (E == 0) return GS_OK;
  if (!A || !B || !G || !target || !xcoms || !ycoms || !pi || !theta || !ok) return fail(c, GS_ERR_ARG, "null pointer");
  return DISPATCH(c, verify(c, ty, E, m, n, A, B, G, target, xcoms, ycoms, pi, theta, ok, true));
}
int gs_prove_statement(gs_ctx* c, int ty, size_t E, int m, int n, const void* X, const void* Y, const void* A,
                       const void* B, const void* G, const void* R, const void* S, const void* T, void* xcoms,
                       void* ycoms, void* pi, void* theta) {
  RC(check_ctx(c, true));
  RC(check_shape(c, ty, m, n));
  if (E == 0) return GS_OK;
  if (!X || !Y || !A || !B || !G || !R || !S || !T || !pi || !theta) return fail(c, GS_ERR_ARG, "null pointer");
  ProveArgs a{ty, E, m, n, X, Y, A, B, G, R, S, T, xcoms, ycoms, pi, theta, true};
  HostPipe hp(c);
  RC(prove_host_stage(c, a, hp));
  RC(prove_host_run(c, a, hp));
  return hp.finish();
}
int gs_verify_statement(gs_ctx* c, int ty, size_t E, int m, int n, const void* A, const void* B, const void* G,
                        const void* target, const void* xcoms, const void* ycoms, const void* pi, const void* theta,
                        uint8_t* ok) {
  RC(check_ctx(c, true));
  RC(check_shape(c, ty, m, n));
  if (E == 0) return GS_OK;
  if (!A || !B || !G || !target || !xcoms || !ycoms || !pi || !theta || !ok) return fail(c, GS_ERR_ARG, "null pointer");
  VerifyArgs a{ty, E, m, n, A, B, G, target, xcoms, ycoms, pi, theta, ok, true};
  HostPipe hp(c);
  RC(verify_host_stage(c, a, hp));
  RC(verify_host_run(c, a, hp));
  return hp.finish();
}

// ---- helpers / hooks -----------------------------------------------------------
int gs_g1_mul_batch_dev(gs_ctx* c, size_t n, const void* p, int bc, const void* k, void* out) {
  RC(check_ctx(c, false));
  if (n == 0) return GS_OK;
  if (!c->endo) {  // plain double-and-add lanes: any curve point (gs_set_option "endo" 0)
    if (c->curve == 0)
      return launch(c, "k_smul_batch.plain.g1", k_smul_batch<Bls12_381, Fq<Bls12_381>, false>, n, 64, n, (const uint8_t*)p,
                    bc, (const Fr<Bls12_381>*)k, (uint8_t*)out);
    return launch(c, "k_smul_batch.plain.g1", k_smul_batch<Bn254, Fq<Bn254>, false>, n, 64, n, (const uint8_t*)p, bc,
                  (const Fr<Bn254>*)k, (uint8_t*)out);
  }
  if (c->curve == 0)
    return launch(c, "k_smul_batch.g1", k_smul_batch<Bls12_381, Fq<Bls12_381>>, n, 64, n, (const uint8_t*)p, bc,
                  (const Fr<Bls12_381>*)k, (uint8_t*)out);
  return launch(c, "k_smul_batch.g1", k_smul_batch<Bn254, Fq<Bn254>>, n, 64, n, (const uint8_t*)p, bc,
                (const Fr<Bn254>*)k, (uint8_t*)out);
}
int gs_g2_mul_batch_dev(gs_ctx* c, size_t n, const void* p, int bc, const void* k, void* out) {
  RC(check_ctx(c, false));
  if (n == 0) return GS_OK;
  if (!c->endo) {  // plain double-and-add lanes: any curve point (gs_set_option "endo" 0)
    if (c->curve == 0)
      return launch(c, "k_smul_batch.plain.g2", k_smul_batch<Bls12_381, Fp2<Bls12_381>, false>, n, 64, n, (const uint8_t*)p,
                    bc, (const Fr<Bls12_381>*)k, (uint8_t*)out);
    return launch(c, "k_smul_batch.plain.g2", k_smul_batch<Bn254, Fp2<Bn254>, false>, n, 64, n, (const uint8_t*)p, bc,
                  (const Fr<Bn254>*)k, (uint8_t*)out);
  }
  if (c->curve == 0)
    return launch(c, "k_smul_batch.g2", k_smul_batch<Bls12_381, Fp2<Bls12_381>>, n, 64, n, (const uint8_t*)p, bc,
                  (const Fr<Bls12_381>*)k, (uint8_t*)out);
  return launch(c, "k_smul_batch.g2", k_smul_batch<Bn254, Fp2<Bn254>>, n, 64, n, (const uint8_t*)p, bc,
                (const Fr<Bn254>*)k, (uint8_t*)out);
}
int gs_g1_mul_batch(gs_ctx* c, size_t n, const void* p, int bc, const void* k, void* out) {
  RC(check_ctx(c, false));
  size_t fq = sz_fq(c->curve);
  HostStage st(c);
  void *dp, *dk, *dout;
  RC(st.in(p, (bc ? 1 : n) * 2 * fq, &dp));
  RC(st.in(k, n * SZ_FR, &dk));
  RC(st.out(out, n * 2 * fq, &dout));
  RC(gs_g1_mul_batch_dev(c, n, dp, bc, dk, dout));
  return st.back(out, dout, n * 2 * fq);
}
int gs_g2_mul_batch(gs_ctx* c, size_t n, const void* p, int bc, const void* k, void* out) {
  RC(check_ctx(c, false));
  size_t fq = sz_fq(c->curve);
  HostStage st(c);
  void *dp, *dk, *dout;
  RC(st.in(p, (bc ? 1 : n) * 4 * fq, &dp));
  RC(st.in(k, n * SZ_FR, &dk));
  RC(st.out(out, n * 4 * fq, &dout));
  RC(gs_g2_mul_batch_dev(c, n, dp, bc, dk, dout));
  return st.back(out, dout, n * 4 * fq);
}

int gs_multi_pairing_batch_dev(gs_ctx* c, size_t n, int k, const void* p, const void* q, void* out) {
  RC(check_ctx(c, false));
  if (n == 0) return GS_OK;
  if (k < 0) return GS_ERR_ARG;
  if (c->curve == 0)
    return launch(c, "k_multi_pairing", k_multi_pairing<Bls12_381>, n, 64, n, k, (const uint8_t*)p,
                  (const uint8_t*)q, (uint8_t*)out);
  return launch(c, "k_multi_pairing", k_multi_pairing<Bn254>, n, 64, n, k, (const uint8_t*)p, (const uint8_t*)q,
                (uint8_t*)out);
}
int gs_multi_pairing_batch(gs_ctx* c, size_t n, int k, const void* p, const void* q, void* out) {
  RC(check_ctx(c, false));
  size_t fq = sz_fq(c->curve);
  HostStage st(c);
  void *dp, *dq, *dout;
  RC(st.in(p, n * k * 2 * fq, &dp));
  RC(st.in(q, n * k * 4 * fq, &dq));
  RC(st.out(out, n * 12 * fq, &dout));
  RC(gs_multi_pairing_batch_dev(c, n, k, dp, dq, dout));
  return st.back(out, dout, n * 12 * fq);
}

int gs_gt_pow_batch_dev(gs_ctx* c, size_t n, const void* base, const void* k, void* out) {
  RC(check_ctx(c, false));
  if (n == 0) return GS_OK;
  if (c->curve == 0)
    return launch(c, "k_gt_pow", k_gt_pow<Bls12_381>, n, 64, n, (const uint8_t*)base, (const Fr<Bls12_381>*)k,
                  (uint8_t*)out);
  return launch(c, "k_gt_pow", k_gt_pow<Bn254>, n, 64, n, (const uint8_t*)base, (const Fr<Bn254>*)k, (uint8_t*)out);
}

// ComT::pairing_sum: four multi-pairings over the component selections (a,b)
int gs_pairing_sum(gs_ctx* c, int k, const void* x, const void* y, void* out) {
  RC(check_ctx(c, false));
  if (k < 0) return GS_ERR_ARG;
  size_t fq = sz_fq(c->curve);
  size_t g1 = 2 * fq, g2 = 4 * fq;
  std::vector<uint8_t> P(4 * (size_t)k * g1 + 1), Q(4 * (size_t)k * g2 + 1);
  const uint8_t* xb = (const uint8_t*)x;
  const uint8_t* yb = (const uint8_t*)y;
  for (int cell = 0; cell < 4; cell++) {
    int a = cell >> 1, b = cell & 1;
    for (int i = 0; i < k; i++) {
      memcpy(&P[((size_t)cell * k + i) * g1], xb + (size_t)i * 2 * g1 + a * g1, g1);
      memcpy(&Q[((size_t)cell * k + i) * g2], yb + (size_t)i * 2 * g2 + b * g2, g2);
    }
  }
  if (k == 0) {  // empty sum: four GT identities
    gs_ctx* cc = c;
    (void)cc;
  }
  return gs_multi_pairing_batch(c, 4, k, P.data(), Q.data(), out);
}

int gs_mat_left_mul_com1(gs_ctx* c, int rows, int k, const void* lhs, const void* col, void* out) {
  RC(check_ctx(c, false));
  if (rows <= 0 || k <= 0) return GS_OK;  // empty product (data_structures.rs:697-702)
  return WIRE_DISPATCH((left_mul_impl<Bls12_381, Fq<Bls12_381>>(c, rows, k, lhs, col, out)),
                       (left_mul_impl<Bn254, Fq<Bn254>>(c, rows, k, lhs, col, out)));
}
int gs_mat_left_mul_com2(gs_ctx* c, int rows, int k, const void* lhs, const void* col, void* out) {
  RC(check_ctx(c, false));
  if (rows <= 0 || k <= 0) return GS_OK;
  return WIRE_DISPATCH((left_mul_impl<Bls12_381, Fp2<Bls12_381>>(c, rows, k, lhs, col, out)),
                       (left_mul_impl<Bn254, Fp2<Bn254>>(c, rows, k, lhs, col, out)));
}

int gs_fr_matmul(gs_ctx* c, int rows, int inner, int cols, const void* lhs, const void* rhs, void* out) {
  RC(check_ctx(c, false));
  if (rows <= 0 || inner <= 0 || cols <= 0) return GS_OK;  // empty product
  if (!lhs || !rhs || !out) return fail(c, GS_ERR_ARG, "null pointer");
  if (rows > 4096 || inner > 4096 || cols > 4096) return fail(c, GS_ERR_SHAPE, "matrix too large");
  return WIRE_DISPATCH(fr_matmul_impl<Bls12_381>(c, rows, inner, cols, lhs, rhs, out),
                       fr_matmul_impl<Bn254>(c, rows, inner, cols, lhs, rhs, out));
}

// ---- batched (RLC) verifier ---------------------------------------------------------
int gs_verify_batch_rlc_dev(gs_ctx* c, int ty, size_t N, int m, int n, const void* A, const void* B, const void* G,
                            const void* target, const void* xcoms, const void* ycoms, const void* pi,
                            const void* theta, const uint64_t* rho, void* acc) {
  RC(check_ctx(c, true));
  RC(check_shape(c, ty, m, n));
  if (!A || !B || !G || !target || !xcoms || !ycoms || !pi || !theta || !rho || !acc || N == 0)
    return fail(c, GS_ERR_ARG, "null pointer or empty batch");
  return DISPATCH(c, verify_rlc(c, ty, N, m, n, A, B, G, target, xcoms, ycoms, pi, theta, rho, acc));
}
int gs_verify_batch_rlc(gs_ctx* c, int ty, size_t N, int m, int n, const void* A, const void* B, const void* G,
                        const void* target, const void* xcoms, const void* ycoms, const void* pi, const void* theta,
                        const uint64_t* rho, void* acc, uint8_t* ok_all) {
  RC(check_ctx(c, true));
  RC(check_shape(c, ty, m, n));
  if (N == 0) return fail(c, GS_ERR_ARG, "empty batch");
  size_t fq = sz_fq(c->curve);
  bool xg = x_is_group(ty), yg = y_is_group(ty);
  int kx = xg ? 2 : 1, ky = yg ? 2 : 1;
  size_t sx = xg ? 2 * fq : SZ_FR, sy = yg ? 4 * fq : SZ_FR;
  size_t st_ = ty == GS_PPE ? 12 * fq : ty == GS_MSMEG1 ? 2 * fq : ty == GS_MSMEG2 ? 4 * fq : SZ_FR;
  if (!rho) return fail(c, GS_ERR_ARG, "null pointer");
  for (size_t i = 0; i < 4 * N; i++)  // a zero exponent drops its cell equation from the combined check
    if (rho[i] == 0) return fail(c, GS_ERR_ARG, "rho must be non-zero (see the rho contract in gs_amd.h)");
  HostStage st(c);
  void *dA, *dB, *dG, *dt, *dxc, *dyc, *dpi, *dth, *drho, *dacc;
  std::vector<uint8_t> hacc(2 * 12 * fq);
  RC(st.in(A, N * n * sx, &dA));
  RC(st.in(B, N * m * sy, &dB));
  RC(st.in(G, N * m * n * SZ_FR, &dG));
  RC(st.in(target, N * st_, &dt));
  RC(st.in(xcoms, N * m * 4 * fq, &dxc));
  RC(st.in(ycoms, N * n * 8 * fq, &dyc));
  RC(st.in(pi, N * kx * 8 * fq, &dpi));
  RC(st.in(theta, N * ky * 4 * fq, &dth));
  RC(st.in(rho, N * 4 * sizeof(uint64_t), &drho));
  RC(st.out(hacc.data(), hacc.size(), &dacc));
  RC(gs_verify_batch_rlc_dev(c, ty, N, m, n, dA, dB, dG, dt, dxc, dyc, dpi, dth, (const uint64_t*)drho, dacc));
  RC(st.back(hacc.data(), dacc, hacc.size()));
  if (acc) memcpy(acc, hacc.data(), hacc.size());
  if (ok_all) RC(gs_gt_finalize(c, 1, hacc.data(), ok_all));
  return GS_OK;
}
int gs_gt_finalize_dev(gs_ctx* c, size_t count, const void* accs_dev, uint8_t* ok) {
  RC(check_ctx(c, false));
  if (!accs_dev || !ok || count == 0) return fail(c, GS_ERR_ARG, "null pointer");
  return DISPATCH(c, gt_finalize(c, count, accs_dev, ok, true));
}
int gs_gt_finalize(gs_ctx* c, size_t count, const void* accs, uint8_t* ok) {
  RC(check_ctx(c, false));
  if (!accs || !ok || count == 0) return fail(c, GS_ERR_ARG, "null pointer");
  return DISPATCH(c, gt_finalize(c, count, accs, ok));
}

// ---- profiling ---------------------------------------------------------------------


int gs_wire_sizes(int curve, size_t out[6]) {
  if ((curve != 0 && curve != 1) || !out) return GS_ERR_ARG;
  size_t fq = sz_fq(curve);
  out[0] = fq;       // G1 compressed
  out[1] = 2 * fq;   // G1 uncompressed
  out[2] = 2 * fq;   // G2 compressed
  out[3] = 4 * fq;   // G2 uncompressed
  out[4] = SZ_FR;    // Fr
  out[5] = 12 * fq;  // GT
  return GS_OK;
}
int gs_wire_encode_g1(gs_ctx* c, size_t n, int compressed, const void* pts, uint8_t* out) {
  RC(check_ctx(c, false));
  return WIRE_DISPATCH((WireImpl<Bls12_381>::enc_pts<Fq<Bls12_381>>(c, n, compressed, pts, out)),
                       (WireImpl<Bn254>::enc_pts<Fq<Bn254>>(c, n, compressed, pts, out)));
}
int gs_wire_encode_g2(gs_ctx* c, size_t n, int compressed, const void* pts, uint8_t* out) {
  RC(check_ctx(c, false));
  return WIRE_DISPATCH((WireImpl<Bls12_381>::enc_pts<Fp2<Bls12_381>>(c, n, compressed, pts, out)),
                       (WireImpl<Bn254>::enc_pts<Fp2<Bn254>>(c, n, compressed, pts, out)));
}
int gs_wire_decode_g1(gs_ctx* c, size_t n, int compressed, int validate, const uint8_t* in, void* pts, uint8_t* ok) {
  RC(check_ctx(c, false));
  return WIRE_DISPATCH((WireImpl<Bls12_381>::dec_pts<Fq<Bls12_381>>(c, n, compressed, validate, in, pts, ok)),
                       (WireImpl<Bn254>::dec_pts<Fq<Bn254>>(c, n, compressed, validate, in, pts, ok)));
}
int gs_wire_decode_g2(gs_ctx* c, size_t n, int compressed, int validate, const uint8_t* in, void* pts, uint8_t* ok) {
  RC(check_ctx(c, false));
  return WIRE_DISPATCH((WireImpl<Bls12_381>::dec_pts<Fp2<Bls12_381>>(c, n, compressed, validate, in, pts, ok)),
                       (WireImpl<Bn254>::dec_pts<Fp2<Bn254>>(c, n, compressed, validate, in, pts, ok)));
}
// ---- subgroup safety at the boundary ------------------------------------------------------
int gs_validate_points_dev(gs_ctx* c, int group, size_t n, const void* pts, uint8_t* ok) {
  RC(check_ctx(c, false));
  if (group != 1 && group != 2) return fail(c, GS_ERR_ARG, "group: 1 (G1) or 2 (G2)");
  if (n == 0) return GS_OK;
  if (!pts || !ok) return fail(c, GS_ERR_ARG, "null pointer");
  if (c->curve == 0)
    return group == 1 ? WireImpl<Bls12_381>::validate_pts_dev<Fq<Bls12_381>>(c, n, pts, ok)
                      : WireImpl<Bls12_381>::validate_pts_dev<Fp2<Bls12_381>>(c, n, pts, ok);
  return group == 1 ? WireImpl<Bn254>::validate_pts_dev<Fq<Bn254>>(c, n, pts, ok)
                    : WireImpl<Bn254>::validate_pts_dev<Fp2<Bn254>>(c, n, pts, ok);
}
int gs_validate_points(gs_ctx* c, int group, size_t n, const void* pts, uint8_t* ok) {
  RC(check_ctx(c, false));
  if (group != 1 && group != 2) return fail(c, GS_ERR_ARG, "group: 1 (G1) or 2 (G2)");
  if (n == 0) return GS_OK;
  if (!pts || !ok) return fail(c, GS_ERR_ARG, "null pointer");
  size_t pb = (group == 1 ? 2 : 4) * sz_fq(c->curve);
  HostStage st(c);
  void *din, *dok;
  RC(st.in(pts, n * pb, &din));
  RC(st.out(ok, n, &dok));
  RC(gs_validate_points_dev(c, group, n, din, (uint8_t*)dok));
  return st.back(ok, dok, n);
}

int gs_wire_encode_fr(gs_ctx* c, size_t n, const void* fr, uint8_t* out) {
  RC(check_ctx(c, false));
  return WIRE_DISPATCH(WireImpl<Bls12_381>::fields(c, 0, 0, 0, n, fr, out, nullptr),
                       WireImpl<Bn254>::fields(c, 0, 0, 0, n, fr, out, nullptr));
}
int gs_wire_decode_fr(gs_ctx* c, size_t n, const uint8_t* in, void* fr, uint8_t* ok) {
  RC(check_ctx(c, false));
  if (!ok) return GS_ERR_ARG;
  return WIRE_DISPATCH(WireImpl<Bls12_381>::fields(c, 0, 1, 0, n, in, fr, ok),
                       WireImpl<Bn254>::fields(c, 0, 1, 0, n, in, fr, ok));
}
int gs_wire_encode_gt(gs_ctx* c, size_t n, const void* gt, uint8_t* out) {
  RC(check_ctx(c, false));
  return WIRE_DISPATCH(WireImpl<Bls12_381>::fields(c, 1, 0, 0, n, gt, out, nullptr),
                       WireImpl<Bn254>::fields(c, 1, 0, 0, n, gt, out, nullptr));
}
int gs_wire_decode_gt(gs_ctx* c, size_t n, int validate, const uint8_t* in, void* gt, uint8_t* ok) {
  RC(check_ctx(c, false));
  if (!ok) return GS_ERR_ARG;
  return WIRE_DISPATCH(WireImpl<Bls12_381>::fields(c, 1, 1, validate, n, in, gt, ok),
                       WireImpl<Bn254>::fields(c, 1, 1, validate, n, in, gt, ok));
}

#if defined(GS_DEBUG_STAMPS)
// diagnosis build only: read and clear the phase stamps of the G1 / G2 Straus lanes (gs_curve.cuh, gs_dbg)
int gs_debug_stamps(gs_ctx* c, unsigned long long* out16) {
  RC(check_ctx(c, false));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  HIPCHK(c, hipMemcpyFromSymbol(out16, HIP_SYMBOL(gs::gs_dbg), 16 * sizeof(unsigned long long)));
  unsigned long long z[16] = {0};
  HIPCHK(c, hipMemcpyToSymbol(HIP_SYMBOL(gs::gs_dbg), z, sizeof z));
  return GS_OK;
}
#endif
int gs_prof_enable(gs_ctx* c, int on) {
  if (!c) return GS_ERR_ARG;
  c->prof = on != 0;
  return GS_OK;
}
int gs_prof_reset(gs_ctx* c) {
  if (!c) return GS_ERR_ARG;
  c->prof_map.clear();
  c->prof_order.clear();
  return GS_OK;
}
int gs_prof_get(gs_ctx* c, int idx, char* name, size_t cap, double* ms, uint64_t* n) {
  if (!c || idx < 0 || (size_t)idx >= c->prof_order.size()) return GS_ERR_ARG;
  const std::string& k = c->prof_order[idx];
  if (name && cap) {
    strncpy(name, k.c_str(), cap - 1);
    name[cap - 1] = 0;
  }
  if (ms) *ms = c->prof_map[k].ms;
  if (n) *n = c->prof_map[k].n;
  return GS_OK;
}
// clock (GHz) the launches under entry idx ran at: sum over their waves of d s_memtime / d s_memrealtime x 100 MHz, taken
// inside those very launches (k_seg stamps); 0 for kernels that are not segmented launches
int gs_prof_get_clock(gs_ctx* c, int idx, double* ghz) {
  if (!c || !ghz || idx < 0 || idx >= (int)c->prof_order.size()) return GS_ERR_ARG;
  const ProfEntry& p = c->prof_map[c->prof_order[idx]];
  *ghz = p.rt > 0 ? p.cyc / p.rt * 0.1 : 0.0;
  return GS_OK;
}
int gs_prof_get_work(gs_ctx* c, int idx, uint64_t* lanes, uint64_t* work) {
  if (!c || idx < 0 || (size_t)idx >= c->prof_order.size()) return GS_ERR_ARG;
  const ProfEntry& p = c->prof_map[c->prof_order[idx]];
  if (lanes) *lanes = p.lanes;
  if (work) *work = p.work;
  return GS_OK;
}

}  // extern "C"
