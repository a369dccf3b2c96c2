// Multi-device host path of the C-ABI (include/wbc_hip.h, wbc_multi_*): ONE host process drives one solver per GPU of
// the node -- the shape of the reference, which is a single C++ process (/root/reference/README.md:58-60), sharded the
// way BASELINE.json's north_star asks ("shards trivially across the 8 GPUs of one node with RCCL over xGMI").
//
// The batch splits into contiguous slices (wbc_shard_range), shard k lives on devices[k] with its own stream; there is
// no data-path collective.  The one optional collective is consumer-side: every device receives all torques, either as
// an RCCL ncclAllGather (communicators from ncclCommInitAll; xGMI is point-to-point, so for 12 words/state this is
// latency- not bandwidth-bound) or as ONE push kernel per shard that writes the shard's block into every device's buffer
// through peer mappings.  RCCL is loaded with dlopen only when that backend is asked for, so single-GPU users of the
// library do not depend on it.
//
// Round 5: the shards are ISSUED IN PARALLEL.  A 4 096-state tick is 13 us of GPU time; issued one shard after the other
// from one host thread (rounds 2-4) eight devices are fed at one launch per ~3 us, i.e. the path was host-bound by
// construction.  Every shard now has a persistent issue thread bound to its device (IssuePool below): an entry point
// validates all shards on the caller's thread, posts ONE ticket, every thread enqueues its shard's part, the caller returns
// when all have.  wbc_solver_options.multi_threads = -1 keeps the serial issue (and a single shard never starts a thread).
//
// Built on the public C-ABI (wbc_solver_create_ex, wbc_step_batch, ...) plus one launcher of launch.hpp (the push kernel).
#include "../../include/wbc_hip.h"

#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstddef>
#include <cstring>
#include <memory>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "host_internal.hpp"
#include "launch.hpp"

using wbc::fail;

#define HIP_TRY(expr)                                                                                     \
  do {                                                                                                    \
    hipError_t e_ = (expr);                                                                               \
    if (e_ != hipSuccess) return fail(WBC_E_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));       \
  } while (0)

namespace {

inline void cpu_relax() {
#if defined(__x86_64__) || defined(__i386__)
  __builtin_ia32_pause();
#elif defined(__aarch64__)
  asm volatile("yield");
#endif
}

struct Rccl {   // the few RCCL entry points the gather needs, resolved at run time
  void* handle = nullptr;
  decltype(&ncclCommInitAll) CommInitAll = nullptr;
  decltype(&ncclCommDestroy) CommDestroy = nullptr;
  decltype(&ncclCommCount) CommCount = nullptr;
  decltype(&ncclAllGather) AllGather = nullptr;
  decltype(&ncclGroupStart) GroupStart = nullptr;
  decltype(&ncclGroupEnd) GroupEnd = nullptr;
  decltype(&ncclGetErrorString) GetErrorString = nullptr;
  bool load(std::string& err) {
    // a process that already carries RCCL (PyTorch's torch.distributed does) shares that copy: same SONAME
    for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
      handle = dlopen(name, RTLD_NOW | RTLD_LOCAL);
      if (handle) break;
    }
    if (!handle) { err = std::string("cannot load RCCL: ") + dlerror(); return false; }
#define RCCL_SYM(f) f = (decltype(f))dlsym(handle, "nccl" #f); if (!f) { err = "RCCL lacks nccl" #f; return false; }
    RCCL_SYM(CommInitAll) RCCL_SYM(CommDestroy) RCCL_SYM(CommCount) RCCL_SYM(AllGather) RCCL_SYM(GroupStart) RCCL_SYM(GroupEnd)
    RCCL_SYM(GetErrorString)
#undef RCCL_SYM
    return true;
  }
};

struct Shard {
  int device = 0;
  wbc_solver* solver = nullptr;
  hipStream_t stream = nullptr;
  hipEvent_t ev = nullptr;        // "my part of the serial gather is enqueued" (wbc_multi_allgather_tau)
  hipStream_t gstream = nullptr;  // overlapped gathers (wbc_multi_allgather_tau_async): the collective runs here, beside the next tick
  hipEvent_t ev_tick = nullptr;   // "everything enqueued on `stream` so far" -- recorded where an overlapped / peer gather is about to order itself behind the ticks
  hipEvent_t ev_slot[2] = {nullptr, nullptr};   // "the gather of buffer slot b is enqueued" (recorded on `gstream`)
  ncclComm_t comm = nullptr;
  void* d_send = nullptr;         // gather staging [nj * cmax] of ragged shards: the gathers on the shard stream ...
  void* d_send_slot[2] = {nullptr, nullptr};   // ... and the overlapped gathers, one per slot (the two forms no longer share one buffer)
  // host-batch convenience: device image of one shard (allocated on first wbc_multi_step_host)
  void* d_host_img = nullptr;
  size_t host_img_cap = 0;        // states
};

struct DeviceScope {
  int prev = -1;
  DeviceScope() { (void)hipGetDevice(&prev); }
  ~DeviceScope() { if (prev >= 0) (void)hipSetDevice(prev); }
};

// ---- persistent issue threads: one per shard, bound to the shard's device.  A job is a reference to a callable on the poster's
// stack: the poster does not return before every thread has run it, so nothing is copied or allocated per tick.
struct JobRef { int (*call)(void*, int) = nullptr; void* ctx = nullptr; };

class IssuePool {
 public:
  struct alignas(64) Worker {
    std::thread th;
    std::atomic<unsigned> seq{0};    // tickets posted
    std::atomic<unsigned> done{0};   // tickets completed
    std::atomic<int> parked{0};
    std::mutex mu;
    std::condition_variable cv;
    int rc = 0;
    std::string err;
  };
  // never throws (it is built under an extern "C" entry point): ok() says whether every worker exists and every thread started; the destructor joins
  // whatever did start (`new Worker` / std::thread construction used to be able to throw through wbc_multi_create, with joinable threads alive)
  IssuePool(const std::vector<int>& devices, int spin_us) noexcept : spin_ns_((long long)spin_us * 1000) {
    try {
      w_.resize(devices.size());
      for (size_t k = 0; k < devices.size(); ++k) w_[k].reset(new Worker);
      for (size_t k = 0; k < devices.size(); ++k) w_[k]->th = std::thread([this, k, dev = devices[k]] { loop((int)k, dev); });
      ok_ = true;
    } catch (...) { ok_ = false; }
  }
  bool ok() const { return ok_; }
  ~IssuePool() {
    quit_.store(true, std::memory_order_seq_cst);
    for (auto& w : w_) if (w && w->th.joinable()) post_one(*w);
    for (auto& w : w_) if (w && w->th.joinable()) w->th.join();
  }
  // runs job(k) on the thread of every shard k and returns the first non-zero status (its message becomes wbc_last_error())
  int run(const JobRef& j) {
    job_ = j;
    for (auto& w : w_) post_one(*w);
    int rc = 0;
    for (auto& w : w_) {
      const unsigned want = w->seq.load(std::memory_order_relaxed);
      while (w->done.load(std::memory_order_acquire) != want) cpu_relax();
      if (w->rc && !rc) rc = fail(w->rc, w->err);
    }
    return rc;
  }
  size_t size() const { return w_.size(); }

 private:
  void post_one(Worker& w) {
    w.seq.fetch_add(1, std::memory_order_seq_cst);
    if (w.parked.load(std::memory_order_seq_cst)) {
      std::lock_guard<std::mutex> lk(w.mu);
      w.cv.notify_one();
    }
  }
  void loop(int k, int dev) {
    const hipError_t bind = dev < 0 ? hipSuccess : hipSetDevice(dev);   // every call this thread makes runs on the shard's device: no device switch per tick (dev < 0: the device-free self-test)
    Worker& w = *w_[(size_t)k];
    unsigned last = 0;
    for (;;) {
      // wait for a ticket: spin while the caller is in a tick loop (a tick is ~13 us), park on the condition variable once it has
      // been quiet for spin_ns_ (a 1 kHz control loop then pays a wake-up per tick instead of a core per shard)
      if (w.seq.load(std::memory_order_acquire) == last) {
        const auto t0 = std::chrono::steady_clock::now();
        for (;;) {
          bool got = false;
          for (int i = 0; i < 64 && !got; ++i) { got = w.seq.load(std::memory_order_acquire) != last; if (!got) cpu_relax(); }
          if (got) break;
          if (std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count() > spin_ns_) {
            std::unique_lock<std::mutex> lk(w.mu);
            w.parked.store(1, std::memory_order_seq_cst);
            w.cv.wait(lk, [&] { return w.seq.load(std::memory_order_seq_cst) != last; });
            w.parked.store(0, std::memory_order_seq_cst);
            break;
          }
        }
      }
      ++last;
      if (quit_.load(std::memory_order_seq_cst)) { w.done.store(last, std::memory_order_release); return; }
      int rc;
      if (bind != hipSuccess) rc = fail(WBC_E_HIP, std::string("issue thread could not bind to its device: ") + hipGetErrorString(bind));   // (never launch on the wrong device)
      else rc = job_.call(job_.ctx, k);
      w.rc = rc;
      if (rc) w.err = wbc_last_error();
      w.done.store(last, std::memory_order_release);
    }
  }
  JobRef job_;
  bool ok_ = false;
  std::atomic<bool> quit_{false};
  long long spin_ns_;
  std::vector<std::unique_ptr<Worker>> w_;
};

}  // namespace

struct wbc_multi {
  int dtype = WBC_F64;
  int nq = 0, nv = 0, nj = 0, nf = 0;
  size_t max_total = 0;
  int backend = WBC_GATHER_NONE;
  int rccl_ranks = 0;
  int observer_order = 0;
  bool push_ok = false;           // peer backend: every device can write every other device's memory (the push kernel); else peer copies
  bool force_copies = false;      // wbc_multi_set_peer_copies: never the push kernel
  // the destination set the push kernel was last cleared for (hipPointerGetAttributes on every tau_all[d]: device memory of devices[d]): checked once per
  // buffer set, not per tick (a buffer that is not plain peer-mapped device memory must take the copy path, not a GPU page fault)
  const void* push_seen[64] = {};
  int push_seen_n = 0;
  bool push_seen_ok = false;
  Rccl rccl;
  std::vector<Shard> sh;
  std::unique_ptr<IssuePool> pool;   // null: the shards are issued one after the other on the caller's thread
  // host time spent inside the tick entry points (wbc_multi_host_stats): what feeding n devices costs the caller
  unsigned long long stat_calls = 0, stat_ns = 0;
  size_t ts() const { return dtype == WBC_F64 ? 8 : 4; }
};

// f(k) for every shard k: on the shards' issue threads (in parallel) or, without them, one after the other here
template <class F> static int for_shards(wbc_multi* mm, F&& f) {
  if (!mm->pool) {
    DeviceScope keep;
    for (int k = 0; k < (int)mm->sh.size(); ++k) {
      hipError_t e = hipSetDevice(mm->sh[(size_t)k].device);
      if (e != hipSuccess) return fail(WBC_E_HIP, std::string("hipSetDevice: ") + hipGetErrorString(e));
      const int rc = f(k);
      if (rc) return rc;
    }
    return WBC_OK;
  }
  using Fn = typename std::remove_reference<F>::type;
  JobRef j;
  j.call = [](void* c, int k) -> int { return (*(Fn*)c)(k); };
  j.ctx = (void*)&f;
  return mm->pool->run(j);
}

struct HostTimer {   // accumulates the caller-visible time of one entry point
  wbc_multi* mm;
  std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
  explicit HostTimer(wbc_multi* m) : mm(m) {}
  ~HostTimer() {
    mm->stat_ns += (unsigned long long)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count();
    ++mm->stat_calls;
  }
};

extern "C" int wbc_shard_range(size_t n_total, int n_shards, int shard, size_t* start, size_t* count) {
  if (n_shards < 1 || shard < 0 || shard >= n_shards || !start || !count) return fail(WBC_E_INVALID, "bad shard arguments");
  const size_t base = n_total / (size_t)n_shards, extra = n_total % (size_t)n_shards;
  *start = (size_t)shard * base + ((size_t)shard < extra ? (size_t)shard : extra);
  *count = base + ((size_t)shard < extra ? 1 : 0);
  return WBC_OK;
}

extern "C" void wbc_multi_destroy(wbc_multi* mm) {
  if (!mm) return;
  mm->pool.reset();   // the issue threads first: nothing is enqueued behind this point
  DeviceScope keep;
  for (Shard& s : mm->sh) {
    (void)hipSetDevice(s.device);
    if (s.stream) (void)hipStreamSynchronize(s.stream);
    if (s.gstream) (void)hipStreamSynchronize(s.gstream);
    if (s.comm && mm->rccl.CommDestroy) (void)mm->rccl.CommDestroy(s.comm);
    if (s.d_send) (void)hipFree(s.d_send);
    for (void* p : s.d_send_slot) if (p) (void)hipFree(p);
    if (s.d_host_img) (void)hipFree(s.d_host_img);
    if (s.ev) (void)hipEventDestroy(s.ev);
    if (s.ev_tick) (void)hipEventDestroy(s.ev_tick);
    for (hipEvent_t e : s.ev_slot) if (e) (void)hipEventDestroy(e);
    if (s.gstream) (void)hipStreamDestroy(s.gstream);
    if (s.stream) (void)hipStreamDestroy(s.stream);
    if (s.solver) wbc_solver_destroy(s.solver);
  }
  // the RCCL library stays loaded: unloading it under a process that may hold other communicators is not safe
  delete mm;
}

extern "C" int wbc_multi_create(const wbc_model* m, const wbc_params* p, int dtype, const int* devices, int n_devices,
                                size_t max_batch_total, int gather_backend, const wbc_solver_options* opt, wbc_multi** out) {
  if (!m || !p || !devices || !out || n_devices < 1 || n_devices > 64 || max_batch_total == 0)
    return fail(WBC_E_INVALID, "bad argument");
  if (gather_backend != WBC_GATHER_NONE && gather_backend != WBC_GATHER_RCCL && gather_backend != WBC_GATHER_PEER_COPY)
    return fail(WBC_E_INVALID, "bad gather backend");
  *out = nullptr;
  if (gather_backend == WBC_GATHER_RCCL)
    for (int i = 0; i < n_devices; ++i)
      for (int j = 0; j < i; ++j)
        if (devices[i] == devices[j])
          return fail(WBC_E_INVALID, "RCCL needs distinct devices (one rank per GPU); use WBC_GATHER_PEER_COPY for shards that share a device");
  // issue threads: the fields live at the END of wbc_solver_options, so a caller built against an older struct gets the defaults
  int multi_threads = 0, multi_spin_us = 200;
  if (opt && opt->struct_size >= offsetof(wbc_solver_options, multi_spin_us) + sizeof(int)) { multi_threads = opt->multi_threads; multi_spin_us = opt->multi_spin_us; }
  if (multi_threads < -1 || multi_threads > 1) return fail(WBC_E_INVALID, "multi_threads must be -1 (serial issue), 0 (auto) or 1 (always)");
  if (multi_spin_us < 0 || multi_spin_us > 1000000) return fail(WBC_E_INVALID, "multi_spin_us must be 0 ... 1000000");
  wbc_multi* mm = new (std::nothrow) wbc_multi;
  if (!mm) return fail(WBC_E_INVALID, "out of memory");
  mm->dtype = dtype; mm->max_total = max_batch_total; mm->backend = gather_backend; mm->observer_order = p->observer_order;
  int rc = wbc_model_dims(m, nullptr, &mm->nq, &mm->nv, &mm->nj, &mm->nf);
  if (rc) { delete mm; return rc; }
  mm->sh.resize((size_t)n_devices);
  DeviceScope keep;
  size_t st0 = 0, cmax = 0;
  (void)wbc_shard_range(max_batch_total, n_devices, 0, &st0, &cmax);   // shard 0 is never smaller than the others
  if (cmax == 0) cmax = 1;
  for (int k = 0; k < n_devices; ++k) {
    Shard& s = mm->sh[(size_t)k];
    s.device = devices[k];
    rc = wbc_solver_create_ex(m, p, dtype, s.device, cmax, opt, &s.solver);
    if (rc) { const std::string keep_msg = wbc_last_error(); wbc_multi_destroy(mm); return fail(rc, "shard " + std::to_string(k) + ": " + keep_msg); }
    hipError_t e = hipSetDevice(s.device);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&s.stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&s.ev, hipEventDisableTiming);
    if (e == hipSuccess && gather_backend != WBC_GATHER_NONE) {
      e = hipStreamCreateWithFlags(&s.gstream, hipStreamNonBlocking);
      if (e == hipSuccess) e = hipEventCreateWithFlags(&s.ev_tick, hipEventDisableTiming);
      for (int b = 0; b < 2 && e == hipSuccess; ++b) e = hipEventCreateWithFlags(&s.ev_slot[b], hipEventDisableTiming);
    }
    if (e == hipSuccess && gather_backend == WBC_GATHER_RCCL) {   // (the staging copies exist for the collective's equal counts only)
      e = hipMalloc(&s.d_send, (size_t)mm->nj * cmax * mm->ts());
      for (int b = 0; b < 2 && e == hipSuccess; ++b) e = hipMalloc(&s.d_send_slot[b], (size_t)mm->nj * cmax * mm->ts());
    }
    if (e != hipSuccess) { wbc_multi_destroy(mm); return fail(WBC_E_HIP, std::string("shard setup: ") + hipGetErrorString(e)); }
  }
  if (gather_backend == WBC_GATHER_PEER_COPY) {
    mm->push_ok = true;
    for (int i = 0; i < n_devices; ++i)
      for (int j = 0; j < n_devices; ++j) {
        if (devices[i] == devices[j]) continue;
        int can = 0;
        (void)hipDeviceCanAccessPeer(&can, devices[i], devices[j]);
        if (can) {
          (void)hipSetDevice(devices[i]);
          const hipError_t e = hipDeviceEnablePeerAccess(devices[j], 0);
          if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) can = 0;
          (void)hipGetLastError();
        }
        if (!can) mm->push_ok = false;   // (without the mapping the copies stage through the host: hipMemcpyPeerAsync below)
      }
  }
  if (gather_backend == WBC_GATHER_RCCL) {
    std::string err;
    if (!mm->rccl.load(err)) { wbc_multi_destroy(mm); return fail(WBC_E_HIP, err); }
    std::vector<ncclComm_t> comms((size_t)n_devices);
    ncclResult_t r = mm->rccl.CommInitAll(comms.data(), n_devices, devices);
    if (r != ncclSuccess) { wbc_multi_destroy(mm); return fail(WBC_E_HIP, std::string("ncclCommInitAll: ") + mm->rccl.GetErrorString(r)); }
    for (int k = 0; k < n_devices; ++k) mm->sh[(size_t)k].comm = comms[(size_t)k];
    int cnt = 0;
    if (mm->rccl.CommCount(comms[0], &cnt) == ncclSuccess) mm->rccl_ranks = cnt;
  }
  int distinct = 0;
  for (int i = 0; i < n_devices; ++i) { bool seen = false; for (int j = 0; j < i; ++j) seen = seen || devices[j] == devices[i]; distinct += seen ? 0 : 1; }
  // Issue threads are OPT-IN (multi_threads = 1).  Round 5 switched them on by itself once the shards sat on more than one device -- the case they are for,
  // and the one case this build has never been able to run: every measurement is from ONE device, where the shards share the runtime's queue lock
  // (bench.py --single-process, profiles/r05a_*: 8 shards 30 -> 17 us per tick call, 2 shards 7.2 -> 10.6 us).  Until a run on several devices exists, auto
  // (0) keeps the serial issue of ABI <= 6; `distinct` is kept for the day the default can move.
  (void)distinct;
  if (multi_threads == 1) {
    std::vector<int> devs(devices, devices + n_devices);
    mm->pool.reset(new (std::nothrow) IssuePool(devs, multi_spin_us));
    if (!mm->pool || !mm->pool->ok()) { mm->pool.reset(); wbc_multi_destroy(mm); return fail(WBC_E_HIP, "could not start the issue threads"); }
  }
  *out = mm;
  return WBC_OK;
}

extern "C" int wbc_multi_size(const wbc_multi* mm) { return mm ? (int)mm->sh.size() : 0; }
extern "C" int wbc_multi_rccl_ranks(const wbc_multi* mm) { return mm ? mm->rccl_ranks : 0; }
extern "C" int wbc_multi_issue_threads(const wbc_multi* mm) { return (mm && mm->pool) ? (int)mm->pool->size() : 0; }
extern "C" wbc_solver* wbc_multi_solver(wbc_multi* mm, int shard) {
  if (!mm || shard < 0 || shard >= (int)mm->sh.size()) return nullptr;
  return mm->sh[(size_t)shard].solver;
}
extern "C" void* wbc_multi_stream(wbc_multi* mm, int shard) {
  if (!mm || shard < 0 || shard >= (int)mm->sh.size()) return nullptr;
  return (void*)mm->sh[(size_t)shard].stream;
}
extern "C" int wbc_multi_device(const wbc_multi* mm, int shard) {
  if (!mm || shard < 0 || shard >= (int)mm->sh.size()) return -1;
  return mm->sh[(size_t)shard].device;
}

extern "C" int wbc_multi_host_stats(wbc_multi* mm, unsigned long long* calls, double* seconds, int reset) {
  if (!mm) return fail(WBC_E_INVALID, "null argument");
  if (calls) *calls = mm->stat_calls;
  if (seconds) *seconds = (double)mm->stat_ns * 1e-9;
  if (reset) { mm->stat_calls = 0; mm->stat_ns = 0; }
  return WBC_OK;
}

// Self-test of the issue threads WITHOUT a device (the CPU test-suite runs it: tests/test_host_abi.py): `threads` threads, `tickets` tickets, every thread must
// run every ticket exactly once and in order; spin_us = 0 makes every wait park on the condition variable (the path a tick loop never takes),
// pause_us > 0 lets the threads park between tickets.  Returns WBC_OK, or WBC_E_INVALID with the first discrepancy in wbc_last_error().
extern "C" int wbc_multi_selftest_issue(int threads, int tickets, int spin_us, int pause_us) {
  if (threads < 1 || threads > 64 || tickets < 1 || spin_us < 0 || pause_us < 0) return fail(WBC_E_INVALID, "bad argument");
  std::vector<int> devs((size_t)threads, -1);   // (no device: the threads bind to none)
  IssuePool pool(devs, spin_us);
  if (!pool.ok()) return fail(WBC_E_HIP, "could not start the issue threads");
  std::vector<long long> seen((size_t)threads, -1), bad((size_t)threads, 0);
  for (int t = 0; t < tickets; ++t) {
    auto job = [&](int k) -> int {
      if (seen[(size_t)k] != t - 1) ++bad[(size_t)k];     // a ticket skipped, or run twice
      seen[(size_t)k] = t;
      return (t % 7 == 3 && k == threads - 1) ? fail(WBC_E_CAPACITY, "ticket " + std::to_string(t)) : WBC_OK;   // an error now and then: it must reach the caller
    };
    JobRef j;
    j.call = [](void* c, int k) -> int { return (*(decltype(job)*)c)(k); };
    j.ctx = (void*)&job;
    const int rc = pool.run(j);
    const int want = (t % 7 == 3) ? WBC_E_CAPACITY : WBC_OK;
    if (rc != want) return fail(WBC_E_INVALID, "ticket " + std::to_string(t) + ": status " + std::to_string(rc) + " instead of " + std::to_string(want));
    if (want && std::string(wbc_last_error()) != "ticket " + std::to_string(t)) return fail(WBC_E_INVALID, "the failing thread's message did not reach the caller");
    if (pause_us) std::this_thread::sleep_for(std::chrono::microseconds(pause_us));
  }
  for (int k = 0; k < threads; ++k)
    if (bad[(size_t)k] || seen[(size_t)k] != tickets - 1) return fail(WBC_E_INVALID, "thread " + std::to_string(k) + " missed or repeated a ticket");
  return WBC_OK;
}

// diagnostics: `iters` empty tickets through the issue threads -- what one for_shards round trip costs the caller apart from the HIP calls inside it
extern "C" int wbc_multi_probe_issue(wbc_multi* mm, int iters, double* seconds) {
  if (!mm || iters < 1 || !seconds) return fail(WBC_E_INVALID, "bad argument");
  const auto t0 = std::chrono::steady_clock::now();
  for (int i = 0; i < iters; ++i) {
    const int rc = for_shards(mm, [](int) { return WBC_OK; });
    if (rc) return rc;
  }
  *seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  return WBC_OK;
}

extern "C" int wbc_multi_set_params(wbc_multi* mm, const wbc_params* p) {
  if (!mm) return fail(WBC_E_INVALID, "null argument");
  int rc = wbc::check_params_public(p);   // checked once, then applied to every shard: the shards never disagree about the parameters
  if (rc) return rc;
  for (Shard& s : mm->sh) { rc = wbc_solver_set_params(s.solver, p); if (rc) return rc; }
  mm->observer_order = p->observer_order;
  return WBC_OK;
}

// every shard's arguments are checked before any shard is enqueued: a bad shard k must not leave shards 0..k-1 one tick ahead
// (their observer state advanced) of the others
static int check_all(wbc_multi* mm, size_t n_total, const wbc_batch_in* in, const wbc_batch_out* out, const wbc_observer_state* obs,
                     bool rollout, int* const* active = nullptr, bool need_active = false) {
  if (n_total > mm->max_total) return fail(WBC_E_CAPACITY, "n_total exceeds max_batch_total");
  const int n = (int)mm->sh.size();
  for (int k = 0; k < n; ++k) {
    size_t st, cnt;
    (void)wbc_shard_range(n_total, n, k, &st, &cnt);
    const int rc = wbc::check_step_args(mm->sh[(size_t)k].solver, cnt, &in[k], &out[k], obs ? &obs[k] : nullptr, rollout);
    if (rc) return rc;
    if (need_active && cnt && !active[k]) return fail(WBC_E_INVALID, "null active-set buffer");
  }
  return WBC_OK;
}

// shard k's tick on its stream (+ the "tick enqueued" event the overlapped gathers wait for)
static int tick_shard(wbc_multi* mm, int k, size_t n_total, const wbc_batch_in* in, const wbc_batch_out* out, const wbc_observer_state* obs,
                      int* const* active, bool record) {
  Shard& s = mm->sh[(size_t)k];
  size_t st, cnt;
  (void)wbc_shard_range(n_total, (int)mm->sh.size(), k, &st, &cnt);
  const int rc = active ? wbc_step_batch_warm(s.solver, cnt, &in[k], &out[k], obs ? &obs[k] : nullptr, active[k], active[k], s.stream)
                        : wbc_step_batch(s.solver, cnt, &in[k], &out[k], obs ? &obs[k] : nullptr, s.stream);
  if (rc) return rc;
  // (only where an overlapped gather follows in the same call: an event record is a release fence in the queue, and two of them per tick pair
  //  on one device cost the ticks more than the launches -- measured, round 5)
  if (record && s.ev_tick) HIP_TRY(hipEventRecord(s.ev_tick, s.stream));
  return WBC_OK;
}

extern "C" int wbc_multi_step_batch(wbc_multi* mm, size_t n_total, const wbc_batch_in* in, const wbc_batch_out* out,
                                    const wbc_observer_state* obs) {
  if (!mm || !in || !out) return fail(WBC_E_INVALID, "null argument");
  HostTimer ht(mm);
  const int rc = check_all(mm, n_total, in, out, obs, false);
  if (rc) return rc;
  return for_shards(mm, [&](int k) { return tick_shard(mm, k, n_total, in, out, obs, nullptr, false); });   // every shard is enqueued before any is looked at: the devices run concurrently
}

// wbc_step_batch_warm per shard: active[k] = shard k's carried active sets (int32 [count_k] on devices[k]), read and rewritten in place
extern "C" int wbc_multi_step_batch_warm(wbc_multi* mm, size_t n_total, const wbc_batch_in* in, const wbc_batch_out* out,
                                         const wbc_observer_state* obs, int* const* active) {
  if (!mm || !in || !out || !active) return fail(WBC_E_INVALID, "null argument");
  HostTimer ht(mm);
  const int rc = check_all(mm, n_total, in, out, obs, false, active, true);
  if (rc) return rc;
  return for_shards(mm, [&](int k) { return tick_shard(mm, k, n_total, in, out, obs, active, false); });
}

extern "C" int wbc_multi_rollout_batch(wbc_multi* mm, size_t n_total, int horizon, const wbc_batch_in* in, const wbc_batch_out* out,
                                       const wbc_observer_state* obs, const void* const* tau_ext) {
  if (!mm || !in || !out) return fail(WBC_E_INVALID, "null argument");
  if (horizon < 1) return fail(WBC_E_INVALID, "horizon must be >= 1");
  HostTimer ht(mm);
  const int rc = check_all(mm, n_total, in, out, obs, true);
  if (rc) return rc;
  const int n = (int)mm->sh.size();
  return for_shards(mm, [&](int k) -> int {   // rank-local for all ticks
    Shard& s = mm->sh[(size_t)k];
    size_t st, cnt;
    (void)wbc_shard_range(n_total, n, k, &st, &cnt);
    const int r = wbc_rollout_batch(s.solver, cnt, horizon, &in[k], &out[k], obs ? &obs[k] : nullptr, tau_ext ? tau_ext[k] : nullptr, nullptr, s.stream);
    return r;
  });
}

// ---- the gather.  Layout of tau_all[d] (on devices[d]): n blocks of nj * count_0 scalars, block j = shard j's [nj][count_j], packed.
struct GatherGeom { size_t cmax, blk; };
static int gather_geom(wbc_multi* mm, size_t n_total, const void* const* tau_local, void* const* tau_all, GatherGeom& g) {
  if (!mm || !tau_local || !tau_all) return fail(WBC_E_INVALID, "null argument");
  if (mm->backend == WBC_GATHER_NONE) return fail(WBC_E_INVALID, "this wbc_multi was created without a gather backend");
  if (n_total > mm->max_total) return fail(WBC_E_CAPACITY, "n_total exceeds max_batch_total");
  const int n = (int)mm->sh.size();
  size_t st0;
  (void)wbc_shard_range(n_total, n, 0, &st0, &g.cmax);
  g.blk = (size_t)mm->nj * g.cmax;
  for (int k = 0; k < n && g.cmax; ++k) if (!tau_local[k] || !tau_all[k]) return fail(WBC_E_INVALID, "null tau buffer");
  return WBC_OK;
}

// shard j's part of a gather on stream `gs` (its shard stream, or its gather stream): RCCL = its rank's ncclAllGather; peer = push my block
// into every device's buffer.  `grouped`: the caller brackets all shards' calls with ncclGroupStart / End (serial issue); the issue threads
// call their rank's collective concurrently instead, the one-thread-per-device form of the library.
// may the push kernel write these destinations?  Every tau_all[d] must be plain device memory of devices[d] (the peer mappings of wbc_multi_create cover
// hipMalloc allocations); anything else -- another device's memory, host memory, a pointer the runtime does not know -- takes the copy path.  Cached per
// buffer set.  (Memory from a virtual-memory pool -- PyTorch expandable segments, hipMallocAsync -- can look like device memory here without being mapped on
// the peers: callers with such buffers switch the push kernel off, wbc_multi_set_peer_copies.)
static bool push_allowed(wbc_multi* mm, void* const* tau_all) {
  if (!mm->push_ok || mm->force_copies) return false;
  const int n = (int)mm->sh.size();
  bool same = mm->push_seen_n == n;
  for (int d = 0; d < n && same; ++d) same = mm->push_seen[d] == tau_all[d];
  if (same) return mm->push_seen_ok;
  bool ok = true;
  for (int d = 0; d < n && ok; ++d) {
    hipPointerAttribute_t at;
    std::memset(&at, 0, sizeof(at));
    const hipError_t e = hipPointerGetAttributes(&at, tau_all[d]);
    if (e != hipSuccess) { (void)hipGetLastError(); ok = false; break; }
    ok = at.type == hipMemoryTypeDevice && at.device == mm->sh[(size_t)d].device;
  }
  for (int d = 0; d < n; ++d) mm->push_seen[d] = tau_all[d];
  mm->push_seen_n = n;
  mm->push_seen_ok = ok;
  return ok;
}

// RCCL, ragged shards: the collective needs equal counts, so a short shard sends from a padded staging copy.  The fallible step IN FRONT of the collective,
// kept apart from it: a rank that fails here must fail before ANY rank has enqueued its ncclAllGather, or the others' streams wait for it for ever
static int gather_pre(wbc_multi* mm, int j, size_t n_total, const GatherGeom& g, const void* const* tau_local, hipStream_t gs, void* staging) {
  if (mm->backend != WBC_GATHER_RCCL) return WBC_OK;
  size_t st, cnt;
  (void)wbc_shard_range(n_total, (int)mm->sh.size(), j, &st, &cnt);
  if (cnt != g.cmax && cnt) HIP_TRY(hipMemcpyAsync(staging, tau_local[j], (size_t)mm->nj * cnt * mm->ts(), hipMemcpyDeviceToDevice, gs));
  return WBC_OK;
}
// ... and the collective (RCCL; gather_pre has run on every shard) or the peer push / copies of shard j
static int gather_post(wbc_multi* mm, int j, size_t n_total, const GatherGeom& g, const void* const* tau_local, void* const* tau_all,
                       hipStream_t gs, void* staging, bool push) {
  const int n = (int)mm->sh.size();
  const size_t ts = mm->ts();
  Shard& src = mm->sh[(size_t)j];
  size_t st, cnt;
  (void)wbc_shard_range(n_total, n, j, &st, &cnt);
  const size_t bytes = (size_t)mm->nj * cnt * ts;
  if (mm->backend == WBC_GATHER_RCCL) {
    const void* send = cnt != g.cmax ? staging : tau_local[j];
    const ncclResult_t r = mm->rccl.AllGather(send, tau_all[j], g.blk, mm->dtype == WBC_F64 ? ncclFloat64 : ncclFloat32, src.comm, gs);
    if (r != ncclSuccess) return fail(WBC_E_HIP, std::string("ncclAllGather: ") + mm->rccl.GetErrorString(r));
    return WBC_OK;
  }
  if (!bytes) return WBC_OK;
  if (push) {   // ONE launch writes my block to all n destinations (peer mappings over xGMI; 2 n^2 -> n runtime calls per gather against the copies)
    void* dst[64];
    int nd = 0;
    for (int d = 0; d < n; ++d) {
      void* p = (char*)tau_all[d] + (size_t)j * g.blk * ts;
      if (p != tau_local[j]) dst[nd++] = p;   // (in-place gather: the tick wrote tau straight into its own block of its own device's buffer)
    }
    if (nd) {
      const hipError_t e = wbc::k_gather_push(gs, tau_local[j], dst, nd, bytes);
      if (e != hipSuccess) return fail(WBC_E_HIP, std::string("gather push launch: ") + hipGetErrorString(e));
    }
    return WBC_OK;
  }
  for (int d = 0; d < n; ++d) {
    char* dst = (char*)tau_all[d] + (size_t)j * g.blk * ts;
    if ((const void*)dst == tau_local[j]) continue;
    if (mm->sh[(size_t)d].device == src.device) HIP_TRY(hipMemcpyAsync(dst, tau_local[j], bytes, hipMemcpyDefault, gs));   // (default kind: a destination the push check refused may be host memory)
    else HIP_TRY(hipMemcpyPeerAsync(dst, mm->sh[(size_t)d].device, tau_local[j], src.device, bytes, gs));
  }
  return WBC_OK;
}

// "everything enqueued on the shard streams so far": what the gathers order themselves behind
static int record_ticks(wbc_multi* mm) {
  return for_shards(mm, [&](int k) -> int { Shard& s = mm->sh[(size_t)k]; HIP_TRY(hipEventRecord(s.ev_tick, s.stream)); return WBC_OK; });
}

// the serial RCCL form of a gather over `streams[k]`: the fallible steps of every rank first, then ONE group call that is ALWAYS closed (an early
// return between ncclGroupStart and ncclGroupEnd used to leave the group open)
template <class StreamOf, class StagingOf>
static int rccl_gather_serial(wbc_multi* mm, size_t n_total, const GatherGeom& g, const void* const* tau_local, void* const* tau_all, StreamOf stream_of, StagingOf staging_of) {
  DeviceScope keep;
  const int n = (int)mm->sh.size();
  for (int k = 0; k < n; ++k) {
    HIP_TRY(hipSetDevice(mm->sh[(size_t)k].device));
    const int rc = gather_pre(mm, k, n_total, g, tau_local, stream_of(k), staging_of(k));
    if (rc) return rc;
  }
  ncclResult_t r = mm->rccl.GroupStart();
  if (r != ncclSuccess) return fail(WBC_E_HIP, std::string("ncclGroupStart: ") + mm->rccl.GetErrorString(r));
  int rc = WBC_OK;
  for (int k = 0; k < n && !rc; ++k) {
    const hipError_t e = hipSetDevice(mm->sh[(size_t)k].device);
    if (e != hipSuccess) { rc = fail(WBC_E_HIP, std::string("hipSetDevice: ") + hipGetErrorString(e)); break; }
    rc = gather_post(mm, k, n_total, g, tau_local, tau_all, stream_of(k), staging_of(k), false);
  }
  r = mm->rccl.GroupEnd();
  if (!rc && r != ncclSuccess) rc = fail(WBC_E_HIP, std::string("ncclGroupEnd: ") + mm->rccl.GetErrorString(r));
  return rc;
}

// behind the tick, on the shard streams; on return every shard stream is ordered behind ALL blocks of its tau_all
extern "C" int wbc_multi_allgather_tau(wbc_multi* mm, size_t n_total, const void* const* tau_local, void* const* tau_all) {
  GatherGeom g;
  int rc = gather_geom(mm, n_total, tau_local, tau_all, g);
  if (rc || g.cmax == 0) return rc;
  HostTimer ht(mm);
  const int n = (int)mm->sh.size();
  if (mm->backend == WBC_GATHER_RCCL) {
    if (mm->pool) {   // issue threads: one ticket for the fallible steps, a second one for the collectives only when every rank got through the first
      rc = for_shards(mm, [&](int k) { Shard& s = mm->sh[(size_t)k]; return gather_pre(mm, k, n_total, g, tau_local, s.stream, s.d_send); });
      if (rc) return rc;
      return for_shards(mm, [&](int k) { Shard& s = mm->sh[(size_t)k]; return gather_post(mm, k, n_total, g, tau_local, tau_all, s.stream, s.d_send, false); });
    }
    return rccl_gather_serial(mm, n_total, g, tau_local, tau_all, [&](int k) { return mm->sh[(size_t)k].stream; }, [&](int k) { return mm->sh[(size_t)k].d_send; });
  }
  rc = record_ticks(mm);
  if (rc) return rc;
  const bool push = push_allowed(mm, tau_all);
  // peer: (1) every shard stream waits for every OTHER shard's tick -- a push lands in buffers that the destination's own tick, or a consumer
  // enqueued behind it, may still be reading -- then pushes its block and records; (2) every shard stream waits for all pushes
  rc = for_shards(mm, [&](int k) -> int {
    Shard& s = mm->sh[(size_t)k];
    for (int d = 0; d < n; ++d)
      if (d != k && mm->sh[(size_t)d].ev_tick) HIP_TRY(hipStreamWaitEvent(s.stream, mm->sh[(size_t)d].ev_tick, 0));
    const int r = gather_post(mm, k, n_total, g, tau_local, tau_all, s.stream, nullptr, push);
    if (r) return r;
    HIP_TRY(hipEventRecord(s.ev, s.stream));
    return WBC_OK;
  });
  if (rc) return rc;
  return for_shards(mm, [&](int k) -> int {
    for (int j = 0; j < n; ++j)
      if (j != k) HIP_TRY(hipStreamWaitEvent(mm->sh[(size_t)k].stream, mm->sh[(size_t)j].ev, 0));
    return WBC_OK;
  });
}

// every shard stream waits (on the device, not on the host) for the last gather of `slot`
static int gather_wait_shard(wbc_multi* mm, int k, int slot) {
  Shard& s = mm->sh[(size_t)k];
  if (mm->backend == WBC_GATHER_RCCL) { HIP_TRY(hipStreamWaitEvent(s.stream, s.ev_slot[slot], 0)); return WBC_OK; }
  // peer: my tau_all receives a block from every shard's gather stream, and my tau is read by my own
  for (Shard& o : mm->sh) HIP_TRY(hipStreamWaitEvent(s.stream, o.ev_slot[slot], 0));
  return WBC_OK;
}

// all shards' parts of the overlapped gather of `slot` (see wbc_multi_allgather_tau_async), each on its GATHER stream, behind the ticks that are enqueued now
// (the ev_tick events have been recorded), beside the next tick
static int gather_async_all(wbc_multi* mm, size_t n_total, const GatherGeom& g, const void* const* tau_local, void* const* tau_all, int slot) {
  const int n = (int)mm->sh.size();
  if (mm->backend == WBC_GATHER_RCCL) {
    // (the collective itself meets the other ranks: a rank's gather stream waits for its own tick only)
    if (!mm->pool) {   // serial issue: one group call over all ranks
      {
        DeviceScope keep;
        for (int k = 0; k < n; ++k) { Shard& s = mm->sh[(size_t)k]; HIP_TRY(hipSetDevice(s.device)); HIP_TRY(hipStreamWaitEvent(s.gstream, s.ev_tick, 0)); }
      }
      const int rc = rccl_gather_serial(mm, n_total, g, tau_local, tau_all, [&](int k) { return mm->sh[(size_t)k].gstream; }, [&](int k) { return mm->sh[(size_t)k].d_send_slot[slot]; });
      if (rc) return rc;
      DeviceScope keep;
      for (int k = 0; k < n; ++k) { Shard& s = mm->sh[(size_t)k]; HIP_TRY(hipSetDevice(s.device)); HIP_TRY(hipEventRecord(s.ev_slot[slot], s.gstream)); }
      return WBC_OK;
    }
    int rc = for_shards(mm, [&](int k) -> int {
      Shard& s = mm->sh[(size_t)k];
      HIP_TRY(hipStreamWaitEvent(s.gstream, s.ev_tick, 0));
      return gather_pre(mm, k, n_total, g, tau_local, s.gstream, s.d_send_slot[slot]);
    });
    if (rc) return rc;
    return for_shards(mm, [&](int k) -> int {
      Shard& s = mm->sh[(size_t)k];
      const int r = gather_post(mm, k, n_total, g, tau_local, tau_all, s.gstream, s.d_send_slot[slot], false);
      if (r) return r;
      HIP_TRY(hipEventRecord(s.ev_slot[slot], s.gstream));
      return WBC_OK;
    });
  }
  const bool push = push_allowed(mm, tau_all);
  return for_shards(mm, [&](int k) -> int {
    Shard& s = mm->sh[(size_t)k];
    // peer pushes write OTHER devices' buffers: behind every shard's tick and whatever read those buffers before it
    for (int d = 0; d < n; ++d) HIP_TRY(hipStreamWaitEvent(s.gstream, mm->sh[(size_t)d].ev_tick, 0));
    const int r = gather_post(mm, k, n_total, g, tau_local, tau_all, s.gstream, s.d_send_slot[slot], push);
    if (r) return r;
    HIP_TRY(hipEventRecord(s.ev_slot[slot], s.gstream));
    return WBC_OK;
  });
}

// 1: the peer gather never uses the push kernel (hipMemcpyPeerAsync / hipMemcpyAsync per block instead) -- for tau_all buffers that are not plain hipMalloc
// memory of their device (virtual-memory pools: PyTorch expandable segments, hipMallocAsync); 0 (default): the push kernel where every destination is
extern "C" int wbc_multi_set_peer_copies(wbc_multi* mm, int on) {
  if (!mm) return fail(WBC_E_INVALID, "null argument");
  mm->force_copies = on != 0;
  return WBC_OK;
}
extern "C" int wbc_multi_gather_pushes(const wbc_multi* mm) { return (mm && mm->backend == WBC_GATHER_PEER_COPY && mm->push_ok && !mm->force_copies && (mm->push_seen_n == 0 || mm->push_seen_ok)) ? 1 : 0; }

// The gather OFF the tick's path: enqueued on the shards' gather streams behind the tick that is on the shard streams now, so that it
// runs beside the NEXT tick.  The caller double-buffers tau (two wbc_batch_out.tau per shard, alternating) and names the buffer's
// slot; before the tick that overwrites a slot's tau it calls wbc_multi_gather_wait(slot).
extern "C" int wbc_multi_allgather_tau_async(wbc_multi* mm, size_t n_total, const void* const* tau_local, void* const* tau_all, int slot) {
  if (!mm) return fail(WBC_E_INVALID, "null argument");
  if (slot < 0 || slot > 1) return fail(WBC_E_INVALID, "slot must be 0 or 1");
  GatherGeom g;
  const int rc = gather_geom(mm, n_total, tau_local, tau_all, g);
  if (rc || g.cmax == 0) return rc;
  HostTimer ht(mm);
  const int rc2 = record_ticks(mm);
  return rc2 ? rc2 : gather_async_all(mm, n_total, g, tau_local, tau_all, slot);
}

extern "C" int wbc_multi_gather_wait(wbc_multi* mm, int slot) {
  if (!mm) return fail(WBC_E_INVALID, "null argument");
  if (slot < 0 || slot > 1) return fail(WBC_E_INVALID, "slot must be 0 or 1");
  if (mm->backend == WBC_GATHER_NONE) return WBC_OK;
  HostTimer ht(mm);
  return for_shards(mm, [&](int k) { return gather_wait_shard(mm, k, slot); });
}

// One call per tick of a double-buffered loop: gather_wait(slot) -> tick writing the slot's tau -> overlapped gather of the slot.  Two
// tickets to the issue threads instead of three calls of the caller (the gather waits for EVERY shard's "tick enqueued" event, so all ticks
// must have been enqueued -- one join between the two halves).
extern "C" int wbc_multi_tick_gather(wbc_multi* mm, size_t n_total, const wbc_batch_in* in, const wbc_batch_out* out, const wbc_observer_state* obs,
                                     int* const* active, void* const* tau_all, int slot) {
  if (!mm || !in || !out || !tau_all) return fail(WBC_E_INVALID, "null argument");
  if (slot < 0 || slot > 1) return fail(WBC_E_INVALID, "slot must be 0 or 1");
  HostTimer ht(mm);
  int rc = check_all(mm, n_total, in, out, obs, false, active, active != nullptr);
  if (rc) return rc;
  const int n = (int)mm->sh.size();
  if (n > 64) return fail(WBC_E_INVALID, "too many shards");
  const void* tau_local[64];
  for (int k = 0; k < n; ++k) tau_local[k] = out[k].tau;
  GatherGeom g;
  rc = gather_geom(mm, n_total, tau_local, tau_all, g);
  if (rc) return rc;
  rc = for_shards(mm, [&](int k) -> int {
    // the tick overwrites out[k].tau, which only shard k's OWN gather stream read (gather k - 2 of this slot): one wait, not one per shard --
    // the pushes of gather k that land in tau_all[k] order themselves behind this tick through ev_tick
    Shard& s = mm->sh[(size_t)k];
    HIP_TRY(hipStreamWaitEvent(s.stream, s.ev_slot[slot], 0));
    return tick_shard(mm, k, n_total, in, out, obs, active, true);
  });
  if (rc || g.cmax == 0) return rc;
  return gather_async_all(mm, n_total, g, tau_local, tau_all, slot);
}

extern "C" int wbc_multi_synchronize(wbc_multi* mm) {
  if (!mm) return fail(WBC_E_INVALID, "null argument");
  return for_shards(mm, [&](int k) -> int {
    Shard& s = mm->sh[(size_t)k];
    HIP_TRY(hipStreamSynchronize(s.stream));
    if (s.gstream) HIP_TRY(hipStreamSynchronize(s.gstream));
    return WBC_OK;
  });
}

// ---- host-resident batch: scatter -> tick -> gather.  Component-major host arrays [ncomp][n_total]; a shard's slice is
// `count` consecutive columns of every component row, i.e. a pitched (2-D) copy.
namespace {
struct HostImg {   // word offsets inside one shard's device image, in units of `cap` columns
  // inputs                                                       outputs
  static constexpr int Q = 0, V = 19, W = 37, A = 43, NRM = 61, MU = 73, TP = 77, FP = 89, IG = 101, R = 119, TAU = 137, F = 149, END = 161;
};
}

extern "C" int wbc_multi_step_host(wbc_multi* mm, size_t n_total, const wbc_batch_in* hin, const wbc_batch_out* hout,
                                   const wbc_observer_state* hobs) {
  if (!mm || !hin || !hout) return fail(WBC_E_INVALID, "null argument");
  if (n_total > mm->max_total) return fail(WBC_E_CAPACITY, "n_total exceeds max_batch_total");
  if (!hin->q || !hin->v || !hin->w_des || !hin->vdot_des || !hin->normals || !hin->mu || !hin->mask || !hout->tau || !hout->f || !hout->status)
    return fail(WBC_E_INVALID, "null host buffer");
  if (hout->M || hout->h || hout->Jc || hout->pf) return fail(WBC_E_INVALID, "the host-batch call returns tau, f, status, iters only");
  if (mm->nq != 19 || mm->nv != 18 || mm->nj != 12 || mm->nf != 4) return fail(WBC_E_TOPOLOGY, "unexpected model dimensions");
  const bool ob = hobs && hobs->integ && hobs->r;
  if (mm->observer_order > 0 && !ob) return fail(WBC_E_INVALID, "observer on: host observer state (integ, r) required");
  if (ob && (!hin->tau_prev || !hin->f_prev)) return fail(WBC_E_INVALID, "observer state given without tau_prev / f_prev");
  const int n = (int)mm->sh.size();
  const size_t ts = mm->ts();
  const int rc = for_shards(mm, [&](int k) -> int {
    size_t st, cnt;
    (void)wbc_shard_range(n_total, n, k, &st, &cnt);
    if (cnt == 0) return WBC_OK;
    Shard& s = mm->sh[(size_t)k];
    if (s.host_img_cap < cnt) {
      if (s.d_host_img) { HIP_TRY(hipStreamSynchronize(s.stream)); HIP_TRY(hipFree(s.d_host_img)); s.d_host_img = nullptr; }
      size_t st0, cap;
      (void)wbc_shard_range(mm->max_total, n, 0, &st0, &cap);
      if (cap < cnt) cap = cnt;
      HIP_TRY(hipMalloc(&s.d_host_img, (size_t)HostImg::END * cap * ts + 3 * cap * sizeof(int)));
      s.host_img_cap = cap;
    }
    char* img = (char*)s.d_host_img;
    const size_t cap = s.host_img_cap;
    auto dptr = [&](int off) { return (void*)(img + (size_t)off * cnt * ts); };   // packed with N = cnt
    int* dints = (int*)(img + (size_t)HostImg::END * cap * ts);
    auto h2d = [&](int off, const void* src, int rows) -> hipError_t {
      return hipMemcpy2DAsync(dptr(off), cnt * ts, (const char*)src + st * ts, n_total * ts, cnt * ts, (size_t)rows, hipMemcpyHostToDevice, s.stream);
    };
    HIP_TRY(h2d(HostImg::Q, hin->q, 19)); HIP_TRY(h2d(HostImg::V, hin->v, 18)); HIP_TRY(h2d(HostImg::W, hin->w_des, 6));
    HIP_TRY(h2d(HostImg::A, hin->vdot_des, 18)); HIP_TRY(h2d(HostImg::NRM, hin->normals, 12)); HIP_TRY(h2d(HostImg::MU, hin->mu, 4));
    HIP_TRY(hipMemcpyAsync(dints, hin->mask + st, cnt * sizeof(int), hipMemcpyHostToDevice, s.stream));
    if (ob) {
      HIP_TRY(h2d(HostImg::TP, hin->tau_prev, 12)); HIP_TRY(h2d(HostImg::FP, hin->f_prev, 12));
      HIP_TRY(h2d(HostImg::IG, hobs->integ, 18)); HIP_TRY(h2d(HostImg::R, hobs->r, 18));
    }
    wbc_batch_in in;
    in.q = dptr(HostImg::Q); in.v = dptr(HostImg::V); in.w_des = dptr(HostImg::W); in.vdot_des = dptr(HostImg::A);
    in.normals = dptr(HostImg::NRM); in.mu = dptr(HostImg::MU); in.mask = dints;
    in.tau_prev = dptr(HostImg::TP); in.f_prev = dptr(HostImg::FP);
    wbc_batch_out out;
    std::memset(&out, 0, sizeof(out));
    out.tau = dptr(HostImg::TAU); out.f = dptr(HostImg::F); out.status = dints + cap; out.iters = dints + 2 * cap;
    wbc_observer_state os{dptr(HostImg::IG), dptr(HostImg::R)};
    const int r = wbc_step_batch(s.solver, cnt, &in, &out, &os, s.stream);
    if (r) return r;
    auto d2h = [&](void* dst, int off, int rows) -> hipError_t {
      return hipMemcpy2DAsync((char*)dst + st * ts, n_total * ts, dptr(off), cnt * ts, cnt * ts, (size_t)rows, hipMemcpyDeviceToHost, s.stream);
    };
    HIP_TRY(d2h(hout->tau, HostImg::TAU, 12)); HIP_TRY(d2h(hout->f, HostImg::F, 12));
    HIP_TRY(hipMemcpyAsync(hout->status + st, dints + cap, cnt * sizeof(int), hipMemcpyDeviceToHost, s.stream));
    if (hout->iters) HIP_TRY(hipMemcpyAsync(hout->iters + st, dints + 2 * cap, cnt * sizeof(int), hipMemcpyDeviceToHost, s.stream));
    if (ob) { HIP_TRY(d2h(hobs->integ, HostImg::IG, 18)); HIP_TRY(d2h(hobs->r, HostImg::R, 18)); }
    return WBC_OK;
  });
  if (rc) return rc;
  return wbc_multi_synchronize(mm);
}
