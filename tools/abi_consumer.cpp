// Stand-alone C++ consumer of the C-ABI for the parity tests (no Python, no torch inside): reads a batch and the
// controller parameters from a binary file the test wrote, runs ONE control tick through the requested entry point
// with raw hipMalloc'ed buffers, and dumps the results for the test to compare with the oracle.
//   abi_consumer <urdf> <in.bin> <out.bin> <mode>
//     single        wbc_solver_create_ex + wbc_step_batch on device 0
//     warm          wbc_step_batch_warm twice: a cold tick on a scratch copy of the batch reports the active sets, the tick that is
//                   dumped starts from them (same inputs: zero iterations everywhere, the same tau / f); prints wbc_plan_tick's answer
//     multi:<k>     wbc_multi_* with k shards dealt round-robin over the visible devices, peer-copy gather of tau,
//                   checked on every device against the shards' own tau
//     rccl          wbc_multi_* with one shard per visible device and the RCCL gather (ncclCommInitAll + ncclAllGather)
//     host:<k>      wbc_multi_step_host with k shards (host-resident batch, pitched scatter / gather)
// in.bin : int64 N, int64 observer_order, 49 doubles of parameters (S6 alpha fn_min fn_max mu_scale dt qp_tol max_iter
//          K1[18] K2[18] -- max_iter as a double), then component-major doubles q[19][N] v[18][N] w_des[6][N]
//          vdot_des[18][N] normals[12][N] mu[4][N] tau_prev[12][N] f_prev[12][N] integ[18][N] r[18][N], then int32 mask[N]
// out.bin: doubles tau[12][N] f[12][N] integ[18][N] r[18][N], int32 status[N] iters[N], int32 rccl_ranks, int32 gather_mismatches
// Build: hipcc -O2 -I include tools/abi_consumer.cpp -L wbc_quadruped_dob_amd/lib -lwbc_hip -Wl,-rpath,$PWD/wbc_quadruped_dob_amd/lib -o tools/abi_consumer.bin
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include "wbc_hip.h"

#define CK(x) do { int rc_ = (x); if (rc_) { std::printf("%s -> %d (%s) %s\n", #x, rc_, wbc_strerror(rc_), wbc_last_error()); return 10; } } while (0)
#define HK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 11; } } while (0)

static const int ROWS[10] = {19, 18, 6, 18, 12, 4, 12, 12, 18, 18};   // q v w a n mu tp fp ig r
enum { Q, V, W, A, NRM, MU, TP, FP, IG, R };

struct Host {
  int64_t N = 0, obs = 0;
  wbc_params prm;
  std::vector<double> in[10];
  std::vector<int> mask;
  std::vector<double> tau, f;
  std::vector<int> status, iters;
};

static bool read_in(const char* path, Host& h) {
  FILE* fp = std::fopen(path, "rb");
  if (!fp) return false;
  double pv[49];
  bool ok = std::fread(&h.N, 8, 1, fp) == 1 && std::fread(&h.obs, 8, 1, fp) == 1 && std::fread(pv, 8, 49, fp) == 49;
  wbc_params_default(&h.prm, WBC_F64);
  for (int i = 0; i < 6; ++i) h.prm.S[i] = pv[i];
  h.prm.alpha = pv[6]; h.prm.fn_min = pv[7]; h.prm.fn_max = pv[8]; h.prm.mu_scale = pv[9]; h.prm.dt = pv[10]; h.prm.qp_tol = pv[11];
  h.prm.max_iter = (int)pv[12]; h.prm.observer_order = (int)h.obs;
  for (int i = 0; i < 18; ++i) { h.prm.K1[i] = pv[13 + i]; h.prm.K2[i] = pv[31 + i]; }
  for (int k = 0; k < 10 && ok; ++k) {
    h.in[k].resize((size_t)ROWS[k] * h.N);
    ok = std::fread(h.in[k].data(), 8, h.in[k].size(), fp) == h.in[k].size();
  }
  h.mask.resize((size_t)h.N);
  ok = ok && std::fread(h.mask.data(), 4, (size_t)h.N, fp) == (size_t)h.N;
  std::fclose(fp);
  h.tau.assign(12 * h.N, 0.0); h.f.assign(12 * h.N, 0.0); h.status.assign(h.N, -1); h.iters.assign(h.N, -1);
  return ok;
}

// device image of one contiguous slice [st, st+cnt) of the host batch, packed component-major with N = cnt
struct DevSlice {
  size_t cnt = 0;
  int device = 0;
  double* in[10] = {};
  double *tau = nullptr, *f = nullptr;
  int *mask = nullptr, *status = nullptr, *iters = nullptr;
};
static int upload(const Host& h, size_t st, size_t cnt, int device, DevSlice& d) {
  d.cnt = cnt; d.device = device;
  HK(hipSetDevice(device));
  const size_t c = cnt ? cnt : 1;
  for (int k = 0; k < 10; ++k) {
    HK(hipMalloc(&d.in[k], (size_t)ROWS[k] * c * 8));
    for (int r = 0; r < ROWS[k] && cnt; ++r)
      HK(hipMemcpy(d.in[k] + (size_t)r * cnt, h.in[k].data() + (size_t)r * h.N + st, cnt * 8, hipMemcpyHostToDevice));
  }
  HK(hipMalloc(&d.tau, 12 * c * 8)); HK(hipMalloc(&d.f, 12 * c * 8));
  HK(hipMemset(d.tau, 0, 12 * c * 8)); HK(hipMemset(d.f, 0, 12 * c * 8));
  HK(hipMalloc(&d.mask, c * 4)); HK(hipMalloc(&d.status, c * 4)); HK(hipMalloc(&d.iters, c * 4));
  if (cnt) HK(hipMemcpy(d.mask, h.mask.data() + st, cnt * 4, hipMemcpyHostToDevice));
  return 0;
}
static int download(Host& h, size_t st, const DevSlice& d) {
  HK(hipSetDevice(d.device));
  const size_t cnt = d.cnt;
  if (!cnt) return 0;
  std::vector<double> t(18 * cnt);
  auto rows = [&](const double* src, std::vector<double>& dst, int nrows) -> int {
    HK(hipMemcpy(t.data(), src, (size_t)nrows * cnt * 8, hipMemcpyDeviceToHost));
    for (int r = 0; r < nrows; ++r) std::memcpy(dst.data() + (size_t)r * h.N + st, t.data() + (size_t)r * cnt, cnt * 8);
    return 0;
  };
  if (rows(d.tau, h.tau, 12) || rows(d.f, h.f, 12) || rows(d.in[IG], h.in[IG], 18) || rows(d.in[R], h.in[R], 18)) return 11;
  HK(hipMemcpy(h.status.data() + st, d.status, cnt * 4, hipMemcpyDeviceToHost));
  HK(hipMemcpy(h.iters.data() + st, d.iters, cnt * 4, hipMemcpyDeviceToHost));
  return 0;
}
static void fill_structs(const Host& h, const DevSlice& d, wbc_batch_in& in, wbc_batch_out& out, wbc_observer_state& os) {
  in = {d.in[Q], d.in[V], d.in[W], d.in[A], d.in[NRM], d.in[MU], d.mask, h.obs ? d.in[TP] : nullptr, h.obs ? d.in[FP] : nullptr};
  std::memset(&out, 0, sizeof(out));
  out.tau = d.tau; out.f = d.f; out.status = d.status; out.iters = d.iters;
  os = {d.in[IG], d.in[R]};
}

int main(int argc, char** argv) {
  if (argc < 5) { std::puts("usage: abi_consumer <urdf> <in.bin> <out.bin> single|warm|multi:<k>|rccl|host:<k>"); return 1; }
  Host h;
  if (!read_in(argv[2], h)) { std::puts("cannot read the input file"); return 2; }
  const std::string mode = argv[4];
  wbc_model* m = nullptr;
  CK(wbc_model_load_urdf(argv[1], nullptr, 0, &m));
  int ndev = 0;
  HK(hipGetDeviceCount(&ndev));
  int rccl_ranks = 0, mismatches = 0;
  const size_t N = (size_t)h.N;
  if (mode == "single") {
    wbc_solver_options opt;
    wbc_solver_options_default(&opt);
    wbc_solver* s = nullptr;
    CK(wbc_solver_create_ex(m, &h.prm, WBC_F64, 0, N, &opt, &s));
    DevSlice d;
    if (upload(h, 0, N, 0, d)) return 11;
    wbc_batch_in in; wbc_batch_out out; wbc_observer_state os;
    fill_structs(h, d, in, out, os);
    hipStream_t st;
    HK(hipStreamCreate(&st));
    CK(wbc_step_batch(s, N, &in, &out, h.obs ? &os : nullptr, st));
    HK(hipStreamSynchronize(st));
    if (download(h, 0, d)) return 11;
    wbc_solver_destroy(s);
  } else if (mode == "warm") {
    wbc_solver* s = nullptr;
    CK(wbc_solver_create(m, &h.prm, WBC_F64, 0, N, &s));
    wbc_tick_plan plan;
    plan.struct_size = sizeof(plan);
    CK(wbc_solver_plan_tick(s, N, 0, 0, 1, &plan));
    DevSlice d1, d2;
    if (upload(h, 0, N, 0, d1) || upload(h, 0, N, 0, d2)) return 11;
    int* active = nullptr;
    HK(hipMalloc(&active, N * sizeof(int)));
    HK(hipMemset(active, 0xFF, N * sizeof(int)));   // garbage: the first tick is told to start cold (active_in = NULL) and overwrites it
    wbc_batch_in in; wbc_batch_out out; wbc_observer_state os;
    hipStream_t st;
    HK(hipStreamCreate(&st));
    fill_structs(h, d1, in, out, os);
    CK(wbc_step_batch_warm(s, N, &in, &out, h.obs ? &os : nullptr, nullptr, active, st));
    fill_structs(h, d2, in, out, os);
    CK(wbc_step_batch_warm(s, N, &in, &out, h.obs ? &os : nullptr, active, active, st));     // in place
    HK(hipStreamSynchronize(st));
    if (download(h, 0, d2)) return 11;
    long long it_sum = 0;
    for (size_t i = 0; i < N; ++i) it_sum += h.iters[i];
    std::printf("warm: plan fused=%d qp=%d, iterations of the warm tick (sum over %zu states) = %lld\n", plan.fused, plan.qp, N, it_sum);
    if (it_sum != 0) ++mismatches;    // the true set of the same problem: no iteration anywhere
    wbc_solver_destroy(s);
  } else if (mode.rfind("host:", 0) == 0) {
    const int k = std::atoi(mode.c_str() + 5);
    std::vector<int> devs((size_t)k);
    for (int i = 0; i < k; ++i) devs[(size_t)i] = i % ndev;
    wbc_multi* mm = nullptr;
    CK(wbc_multi_create(m, &h.prm, WBC_F64, devs.data(), k, N, WBC_GATHER_NONE, nullptr, &mm));
    wbc_batch_in in = {h.in[Q].data(), h.in[V].data(), h.in[W].data(), h.in[A].data(), h.in[NRM].data(), h.in[MU].data(), h.mask.data(),
                       h.obs ? h.in[TP].data() : nullptr, h.obs ? h.in[FP].data() : nullptr};
    wbc_batch_out out;
    std::memset(&out, 0, sizeof(out));
    out.tau = h.tau.data(); out.f = h.f.data(); out.status = h.status.data(); out.iters = h.iters.data();
    wbc_observer_state os = {h.in[IG].data(), h.in[R].data()};
    CK(wbc_multi_step_host(mm, N, &in, &out, h.obs ? &os : nullptr));
    wbc_multi_destroy(mm);
  } else if (mode.rfind("multi:", 0) == 0 || mode == "rccl") {
    const bool rccl = mode == "rccl";
    const int k = rccl ? ndev : std::atoi(mode.c_str() + 6);
    std::vector<int> devs((size_t)k);
    for (int i = 0; i < k; ++i) devs[(size_t)i] = i % ndev;
    wbc_multi* mm = nullptr;
    CK(wbc_multi_create(m, &h.prm, WBC_F64, devs.data(), k, N, rccl ? WBC_GATHER_RCCL : WBC_GATHER_PEER_COPY, nullptr, &mm));
    rccl_ranks = wbc_multi_rccl_ranks(mm);
    std::vector<DevSlice> d((size_t)k);
    std::vector<wbc_batch_in> in((size_t)k);
    std::vector<wbc_batch_out> out((size_t)k);
    std::vector<wbc_observer_state> os((size_t)k);
    std::vector<size_t> st((size_t)k), cnt((size_t)k);
    for (int i = 0; i < k; ++i) {
      CK(wbc_shard_range(N, k, i, &st[(size_t)i], &cnt[(size_t)i]));
      if (upload(h, st[(size_t)i], cnt[(size_t)i], devs[(size_t)i], d[(size_t)i])) return 11;
      fill_structs(h, d[(size_t)i], in[(size_t)i], out[(size_t)i], os[(size_t)i]);
    }
    CK(wbc_multi_step_batch(mm, N, in.data(), out.data(), h.obs ? os.data() : nullptr));
    // every device receives all torques
    const size_t blk = 12 * cnt[0];
    std::vector<const void*> loc((size_t)k);
    std::vector<void*> all((size_t)k);
    for (int i = 0; i < k; ++i) {
      HK(hipSetDevice(devs[(size_t)i]));
      loc[(size_t)i] = d[(size_t)i].tau;
      HK(hipMalloc(&all[(size_t)i], (size_t)k * (blk ? blk : 1) * 8));
      HK(hipMemset(all[(size_t)i], 0xFF, (size_t)k * (blk ? blk : 1) * 8));
    }
    CK(wbc_multi_allgather_tau(mm, N, loc.data(), all.data()));
    CK(wbc_multi_synchronize(mm));
    for (int i = 0; i < k; ++i) if (download(h, st[(size_t)i], d[(size_t)i])) return 11;
    for (int dv = 0; dv < k; ++dv) {   // block j on device dv == shard j's tau, bit for bit
      std::vector<double> got((size_t)k * blk);
      HK(hipSetDevice(devs[(size_t)dv]));
      if (blk) HK(hipMemcpy(got.data(), all[(size_t)dv], got.size() * 8, hipMemcpyDeviceToHost));
      for (int j = 0; j < k; ++j)
        for (int r = 0; r < 12; ++r)
          for (size_t c = 0; c < cnt[(size_t)j]; ++c)
            if (got[(size_t)j * blk + (size_t)r * cnt[(size_t)j] + c] != h.tau[(size_t)r * N + st[(size_t)j] + c]) ++mismatches;
    }
    wbc_multi_destroy(mm);
  } else {
    std::puts("unknown mode");
    return 1;
  }
  wbc_model_free(m);
  FILE* fo = std::fopen(argv[3], "wb");
  if (!fo) return 3;
  std::fwrite(h.tau.data(), 8, h.tau.size(), fo); std::fwrite(h.f.data(), 8, h.f.size(), fo);
  std::fwrite(h.in[IG].data(), 8, h.in[IG].size(), fo); std::fwrite(h.in[R].data(), 8, h.in[R].size(), fo);
  std::fwrite(h.status.data(), 4, h.status.size(), fo); std::fwrite(h.iters.data(), 4, h.iters.size(), fo);
  std::fwrite(&rccl_ranks, 4, 1, fo); std::fwrite(&mismatches, 4, 1, fo);
  std::fclose(fo);
  std::printf("%s: N=%zu devices=%d rccl_ranks=%d gather_mismatches=%d\n", mode.c_str(), N, ndev, rccl_ranks, mismatches);
  return mismatches ? 4 : 0;
}
