// Multi-device host path of the C-ABI (include/wbc_hip.h, wbc_multi_*): ONE host process drives one solver per GPU of
// the node -- the shape of the reference, which is a single C++ process (/root/reference/README.md:58-60), sharded the
// way BASELINE.json's north_star asks ("shards trivially across the 8 GPUs of one node with RCCL over xGMI").
//
// The batch splits into contiguous slices (wbc_shard_range), shard k lives on devices[k] with its own stream; there is
// no data-path collective.  The one optional collective is consumer-side: every device receives all torques, either as
// an RCCL ncclAllGather (communicators from ncclCommInitAll, one group call over all devices; xGMI is point-to-point,
// so for 12 words/state this is latency- not bandwidth-bound) or as peer copies on the shard streams.  RCCL is loaded
// with dlopen only when that backend is asked for, so single-GPU users of the library do not depend on it.
//
// Built on the public C-ABI only (wbc_solver_create_ex, wbc_step_batch, ...): pure host code.
#include "../../include/wbc_hip.h"

#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "host_internal.hpp"

using wbc::fail;

#define HIP_TRY(expr)                                                                                     \
  do {                                                                                                    \
    hipError_t e_ = (expr);                                                                               \
    if (e_ != hipSuccess) return fail(WBC_E_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));       \
  } while (0)

namespace {

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
  hipEvent_t ev = nullptr;        // "my part of the gather / tick is enqueued"
  hipStream_t gstream = nullptr;  // overlapped gathers (wbc_multi_allgather_tau_async): the collective runs here, beside the next tick
  hipEvent_t ev_tick = nullptr;   // "the tick whose tau is to be gathered is enqueued" (recorded on `stream`, waited for by `gstream`)
  hipEvent_t ev_slot[2] = {nullptr, nullptr};   // "the gather of buffer slot b is enqueued" (recorded on `gstream`)
  ncclComm_t comm = nullptr;
  void* d_send = nullptr;         // gather staging [nj * cmax] (ragged shards only)
  // host-batch convenience: device image of one shard (allocated on first wbc_multi_step_host)
  void* d_host_img = nullptr;
  size_t host_img_cap = 0;        // states
};

struct DeviceScope {
  int prev = -1;
  DeviceScope() { (void)hipGetDevice(&prev); }
  ~DeviceScope() { if (prev >= 0) (void)hipSetDevice(prev); }
};

}  // namespace

struct wbc_multi {
  int dtype = WBC_F64;
  int nq = 0, nv = 0, nj = 0, nf = 0;
  size_t max_total = 0;
  int backend = WBC_GATHER_NONE;
  int rccl_ranks = 0;
  int observer_order = 0;
  Rccl rccl;
  std::vector<Shard> sh;
  size_t ts() const { return dtype == WBC_F64 ? 8 : 4; }
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
  DeviceScope keep;
  for (Shard& s : mm->sh) {
    (void)hipSetDevice(s.device);
    if (s.stream) (void)hipStreamSynchronize(s.stream);
    if (s.comm && mm->rccl.CommDestroy) (void)mm->rccl.CommDestroy(s.comm);
    if (s.d_send) (void)hipFree(s.d_send);
    if (s.d_host_img) (void)hipFree(s.d_host_img);
    if (s.gstream) (void)hipStreamSynchronize(s.gstream);
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
    if (e == hipSuccess && gather_backend != WBC_GATHER_NONE) e = hipMalloc(&s.d_send, (size_t)mm->nj * cmax * mm->ts());
    if (e != hipSuccess) { wbc_multi_destroy(mm); return fail(WBC_E_HIP, std::string("shard setup: ") + hipGetErrorString(e)); }
  }
  if (gather_backend == WBC_GATHER_PEER_COPY) {
    for (int i = 0; i < n_devices; ++i)
      for (int j = 0; j < n_devices; ++j) {
        if (devices[i] == devices[j]) continue;
        int can = 0;
        (void)hipDeviceCanAccessPeer(&can, devices[i], devices[j]);
        if (can) { (void)hipSetDevice(devices[i]); hipError_t e = hipDeviceEnablePeerAccess(devices[j], 0); (void)e; (void)hipGetLastError(); }  // (already enabled is fine; without it the copies stage through the host)
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
  *out = mm;
  return WBC_OK;
}

extern "C" int wbc_multi_size(const wbc_multi* mm) { return mm ? (int)mm->sh.size() : 0; }
extern "C" int wbc_multi_rccl_ranks(const wbc_multi* mm) { return mm ? mm->rccl_ranks : 0; }
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

extern "C" int wbc_multi_set_params(wbc_multi* mm, const wbc_params* p) {
  if (!mm) return fail(WBC_E_INVALID, "null argument");
  int rc = wbc::check_params_public(p);   // checked once, then applied to every shard: the shards never disagree about the parameters
  if (rc) return rc;
  for (Shard& s : mm->sh) { rc = wbc_solver_set_params(s.solver, p); if (rc) return rc; }
  mm->observer_order = p->observer_order;
  return WBC_OK;
}

extern "C" int wbc_multi_step_batch(wbc_multi* mm, size_t n_total, const wbc_batch_in* in, const wbc_batch_out* out,
                                    const wbc_observer_state* obs) {
  if (!mm || !in || !out) return fail(WBC_E_INVALID, "null argument");
  if (n_total > mm->max_total) return fail(WBC_E_CAPACITY, "n_total exceeds max_batch_total");
  const int n = (int)mm->sh.size();
  for (int k = 0; k < n; ++k) {   // every shard's arguments are checked before any shard is enqueued: a bad shard k must not
    size_t st, cnt;               // leave shards 0..k-1 one tick ahead (their observer state advanced) of the others
    (void)wbc_shard_range(n_total, n, k, &st, &cnt);
    int rc = wbc::check_step_args(mm->sh[(size_t)k].solver, cnt, &in[k], &out[k], obs ? &obs[k] : nullptr, false);
    if (rc) return rc;
  }
  for (int k = 0; k < n; ++k) {   // enqueue every shard before looking at any: the devices run concurrently
    size_t st, cnt;
    (void)wbc_shard_range(n_total, n, k, &st, &cnt);
    int rc = wbc_step_batch(mm->sh[(size_t)k].solver, cnt, &in[k], &out[k], obs ? &obs[k] : nullptr, mm->sh[(size_t)k].stream);
    if (rc) return rc;
  }
  return WBC_OK;
}

// wbc_step_batch_warm per shard: active[k] = shard k's carried active sets (int32 [count_k] on devices[k]), read and rewritten in place
extern "C" int wbc_multi_step_batch_warm(wbc_multi* mm, size_t n_total, const wbc_batch_in* in, const wbc_batch_out* out,
                                         const wbc_observer_state* obs, int* const* active) {
  if (!mm || !in || !out || !active) return fail(WBC_E_INVALID, "null argument");
  if (n_total > mm->max_total) return fail(WBC_E_CAPACITY, "n_total exceeds max_batch_total");
  const int n = (int)mm->sh.size();
  for (int k = 0; k < n; ++k) {   // validate all shards first (see wbc_multi_step_batch)
    size_t st, cnt;
    (void)wbc_shard_range(n_total, n, k, &st, &cnt);
    int rc = wbc::check_step_args(mm->sh[(size_t)k].solver, cnt, &in[k], &out[k], obs ? &obs[k] : nullptr, false);
    if (rc) return rc;
    if (cnt && !active[k]) return fail(WBC_E_INVALID, "null active-set buffer");
  }
  for (int k = 0; k < n; ++k) {
    size_t st, cnt;
    (void)wbc_shard_range(n_total, n, k, &st, &cnt);
    int rc = wbc_step_batch_warm(mm->sh[(size_t)k].solver, cnt, &in[k], &out[k], obs ? &obs[k] : nullptr, active[k], active[k],
                                 mm->sh[(size_t)k].stream);
    if (rc) return rc;
  }
  return WBC_OK;
}

extern "C" int wbc_multi_rollout_batch(wbc_multi* mm, size_t n_total, int horizon, const wbc_batch_in* in, const wbc_batch_out* out,
                                       const wbc_observer_state* obs, const void* const* tau_ext) {
  if (!mm || !in || !out) return fail(WBC_E_INVALID, "null argument");
  if (n_total > mm->max_total) return fail(WBC_E_CAPACITY, "n_total exceeds max_batch_total");
  const int n = (int)mm->sh.size();
  if (horizon < 1) return fail(WBC_E_INVALID, "horizon must be >= 1");
  for (int k = 0; k < n; ++k) {   // validate all shards first (see wbc_multi_step_batch)
    size_t st, cnt;
    (void)wbc_shard_range(n_total, n, k, &st, &cnt);
    int rc = wbc::check_step_args(mm->sh[(size_t)k].solver, cnt, &in[k], &out[k], obs ? &obs[k] : nullptr, true);
    if (rc) return rc;
  }
  for (int k = 0; k < n; ++k) {   // rank-local for all ticks
    size_t st, cnt;
    (void)wbc_shard_range(n_total, n, k, &st, &cnt);
    int rc = wbc_rollout_batch(mm->sh[(size_t)k].solver, cnt, horizon, &in[k], &out[k], obs ? &obs[k] : nullptr,
                               tau_ext ? tau_ext[k] : nullptr, nullptr, mm->sh[(size_t)k].stream);
    if (rc) return rc;
  }
  return WBC_OK;
}

// the gather on the shard streams (side = false: behind the tick) or on the shards' gather streams (side = true: beside the next tick)
static int gather_impl(wbc_multi* mm, size_t n_total, const void* const* tau_local, void* const* tau_all, bool side) {
#define GSTREAM(sh_) (side ? (sh_).gstream : (sh_).stream)
  if (!mm || !tau_local || !tau_all) return fail(WBC_E_INVALID, "null argument");
  if (mm->backend == WBC_GATHER_NONE) return fail(WBC_E_INVALID, "this wbc_multi was created without a gather backend");
  if (n_total > mm->max_total) return fail(WBC_E_CAPACITY, "n_total exceeds max_batch_total");
  const int n = (int)mm->sh.size();
  const size_t ts = mm->ts();
  size_t st0, cmax;
  (void)wbc_shard_range(n_total, n, 0, &st0, &cmax);
  if (cmax == 0) return WBC_OK;
  const size_t blk = (size_t)mm->nj * cmax;   // elements per block of tau_all: shard j's [nj][count_j], packed, then padding
  DeviceScope keep;
  if (mm->backend == WBC_GATHER_RCCL) {
    std::vector<const void*> send((size_t)n);
    for (int k = 0; k < n; ++k) {
      size_t st, cnt;
      (void)wbc_shard_range(n_total, n, k, &st, &cnt);
      if (!tau_local[k] || !tau_all[k]) return fail(WBC_E_INVALID, "null tau buffer");
      send[(size_t)k] = tau_local[k];
      if (cnt != cmax) {   // ragged: the collective needs equal counts, so the short shards send from a padded staging copy
        Shard& s = mm->sh[(size_t)k];
        HIP_TRY(hipSetDevice(s.device));
        if (cnt) HIP_TRY(hipMemcpyAsync(s.d_send, tau_local[k], (size_t)mm->nj * cnt * ts, hipMemcpyDeviceToDevice, GSTREAM(s)));
        send[(size_t)k] = s.d_send;
      }
    }
    ncclResult_t r = mm->rccl.GroupStart();
    for (int k = 0; k < n && r == ncclSuccess; ++k) {
      Shard& s = mm->sh[(size_t)k];
      r = mm->rccl.AllGather(send[(size_t)k], tau_all[k], blk, mm->dtype == WBC_F64 ? ncclFloat64 : ncclFloat32, s.comm, GSTREAM(s));
    }
    const ncclResult_t r2 = mm->rccl.GroupEnd();
    if (r == ncclSuccess) r = r2;
    if (r != ncclSuccess) return fail(WBC_E_HIP, std::string("ncclAllGather: ") + mm->rccl.GetErrorString(r));
    return WBC_OK;
  }
  // peer copies: shard j pushes its block to every device on ITS stream (behind its tick), then every stream waits for all pushes
  for (int j = 0; j < n; ++j) {
    size_t st, cnt;
    (void)wbc_shard_range(n_total, n, j, &st, &cnt);
    Shard& src = mm->sh[(size_t)j];
    if (!tau_local[j] || !tau_all[j]) return fail(WBC_E_INVALID, "null tau buffer");
    HIP_TRY(hipSetDevice(src.device));
    const size_t bytes = (size_t)mm->nj * cnt * ts;
    for (int d = 0; d < n && bytes; ++d) {
      char* dst = (char*)tau_all[d] + (size_t)j * blk * ts;
      if (mm->sh[(size_t)d].device == src.device) HIP_TRY(hipMemcpyAsync(dst, tau_local[j], bytes, hipMemcpyDeviceToDevice, GSTREAM(src)));
      else HIP_TRY(hipMemcpyPeerAsync(dst, mm->sh[(size_t)d].device, tau_local[j], src.device, bytes, GSTREAM(src)));
    }
    HIP_TRY(hipEventRecord(src.ev, GSTREAM(src)));
  }
  for (int d = 0; d < n; ++d) {
    HIP_TRY(hipSetDevice(mm->sh[(size_t)d].device));
    for (int j = 0; j < n; ++j)
      if (j != d) HIP_TRY(hipStreamWaitEvent(GSTREAM(mm->sh[(size_t)d]), mm->sh[(size_t)j].ev, 0));
  }
  return WBC_OK;
#undef GSTREAM
}

extern "C" int wbc_multi_allgather_tau(wbc_multi* mm, size_t n_total, const void* const* tau_local, void* const* tau_all) {
  return gather_impl(mm, n_total, tau_local, tau_all, false);
}

// The gather OFF the tick's path: enqueued on the shards' gather streams behind the tick that is on the shard streams now, so that it
// runs beside the NEXT tick.  The caller double-buffers tau (two wbc_batch_out.tau per shard, alternating) and names the buffer's
// slot; before the tick that overwrites a slot's tau it calls wbc_multi_gather_wait(slot).
extern "C" int wbc_multi_allgather_tau_async(wbc_multi* mm, size_t n_total, const void* const* tau_local, void* const* tau_all, int slot) {
  if (!mm) return fail(WBC_E_INVALID, "null argument");
  if (slot < 0 || slot > 1) return fail(WBC_E_INVALID, "slot must be 0 or 1");
  if (mm->backend == WBC_GATHER_NONE) return fail(WBC_E_INVALID, "this wbc_multi was created without a gather backend");
  {
    DeviceScope keep;
    for (Shard& s : mm->sh) {   // every gather stream waits for ITS shard's tick (peer copies read only the source shard's tau)
      HIP_TRY(hipSetDevice(s.device));
      HIP_TRY(hipEventRecord(s.ev_tick, s.stream));
      HIP_TRY(hipStreamWaitEvent(s.gstream, s.ev_tick, 0));
    }
  }
  const int rc = gather_impl(mm, n_total, tau_local, tau_all, true);
  if (rc) return rc;
  DeviceScope keep;
  for (Shard& s : mm->sh) {
    HIP_TRY(hipSetDevice(s.device));
    HIP_TRY(hipEventRecord(s.ev_slot[slot], s.gstream));
  }
  return WBC_OK;
}

// every shard stream waits (on the device, not on the host) for the last gather of `slot`: call it before the tick that overwrites
// that slot's tau, and before reading that slot's tau_all on a shard stream
extern "C" int wbc_multi_gather_wait(wbc_multi* mm, int slot) {
  if (!mm) return fail(WBC_E_INVALID, "null argument");
  if (slot < 0 || slot > 1) return fail(WBC_E_INVALID, "slot must be 0 or 1");
  if (mm->backend == WBC_GATHER_NONE) return WBC_OK;
  DeviceScope keep;
  for (Shard& s : mm->sh) {
    HIP_TRY(hipSetDevice(s.device));
    HIP_TRY(hipStreamWaitEvent(s.stream, s.ev_slot[slot], 0));
  }
  return WBC_OK;
}

extern "C" int wbc_multi_synchronize(wbc_multi* mm) {
  if (!mm) return fail(WBC_E_INVALID, "null argument");
  DeviceScope keep;
  for (Shard& s : mm->sh) {
    HIP_TRY(hipSetDevice(s.device));
    HIP_TRY(hipStreamSynchronize(s.stream));
    if (s.gstream) HIP_TRY(hipStreamSynchronize(s.gstream));
  }
  return WBC_OK;
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
  DeviceScope keep;
  for (int k = 0; k < n; ++k) {
    size_t st, cnt;
    (void)wbc_shard_range(n_total, n, k, &st, &cnt);
    if (cnt == 0) continue;
    Shard& s = mm->sh[(size_t)k];
    HIP_TRY(hipSetDevice(s.device));
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
    int rc = wbc_step_batch(s.solver, cnt, &in, &out, &os, s.stream);
    if (rc) return rc;
    auto d2h = [&](void* dst, int off, int rows) -> hipError_t {
      return hipMemcpy2DAsync((char*)dst + st * ts, n_total * ts, dptr(off), cnt * ts, cnt * ts, (size_t)rows, hipMemcpyDeviceToHost, s.stream);
    };
    HIP_TRY(d2h(hout->tau, HostImg::TAU, 12)); HIP_TRY(d2h(hout->f, HostImg::F, 12));
    HIP_TRY(hipMemcpyAsync(hout->status + st, dints + cap, cnt * sizeof(int), hipMemcpyDeviceToHost, s.stream));
    if (hout->iters) HIP_TRY(hipMemcpyAsync(hout->iters + st, dints + 2 * cap, cnt * sizeof(int), hipMemcpyDeviceToHost, s.stream));
    if (ob) { HIP_TRY(d2h(hobs->integ, HostImg::IG, 18)); HIP_TRY(d2h(hobs->r, HostImg::R, 18)); }
  }
  return wbc_multi_synchronize(mm);
}
