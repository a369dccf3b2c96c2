// Stand-alone C++ consumer of the C-ABI (no Python, no torch): what a C++ host such as the reference's
// ROS node would do.  Loads the URDF, runs one single-robot tick through wbc_compute_torques and a
// 1024-state batch through wbc_step_batch with raw hipMalloc'ed buffers.  Build:
//   hipcc -O2 -I include tools/abi_smoke.cpp -L wbc_quadruped_dob_amd/lib -lwbc_hip -Wl,-rpath,$PWD/wbc_quadruped_dob_amd/lib -o /tmp/abi_smoke
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>
#include "wbc_hip.h"
#include "wbc/quadruped_wbc.hpp"   // header-only C++ host class over the same ABI

#define CK(x) do { int rc_ = (x); if (rc_) { std::printf("%s -> %d (%s) %s\n", #x, rc_, wbc_strerror(rc_), wbc_last_error()); return 1; } } while (0)

int main(int argc, char** argv) {
  const char* urdf = argc > 1 ? argv[1] : "wbc_quadruped_dob_amd/assets/synthetic_quadruped.urdf";
  wbc_model* m = nullptr;
  CK(wbc_model_load_urdf(urdf, nullptr, 0, &m));
  int nb, nq, nv, nj, nf;
  CK(wbc_model_dims(m, &nb, &nq, &nv, &nj, &nf));
  std::printf("model: nb=%d nq=%d nv=%d nj=%d nf=%d mass=%.3f\n", nb, nq, nv, nj, nf, wbc_model_total_mass(m));
  wbc_params p;
  wbc_params_default(&p, WBC_F64);
  wbc_solver* s = nullptr;
  const size_t N = 1024;
  CK(wbc_solver_create(m, &p, WBC_F64, 0, N, &s));
  // single robot, standing still: sum of GRFs must carry the weight
  double q[19] = {0, 0, 0.4, 0, 0, 0, 1}, v[18] = {0}, w[6] = {0, 0, wbc_model_total_mass(m) * 9.81, 0, 0, 0}, a[18] = {0};
  for (int l = 0; l < 4; ++l) { q[7 + 3 * l] = 0.05; q[8 + 3 * l] = 0.75; q[9 + 3 * l] = -1.5; }
  double nrm[12], mu[4] = {0.6, 0.6, 0.6, 0.6}, tau[12], f[12];
  for (int k = 0; k < 4; ++k) { nrm[3 * k] = 0; nrm[3 * k + 1] = 0; nrm[3 * k + 2] = 1; }
  int st = -1;
  CK(wbc_compute_torques(s, q, v, w, a, nrm, mu, 0xF, nullptr, nullptr, nullptr, nullptr, tau, f, &st));
  double fz = f[2] + f[5] + f[8] + f[11];
  std::printf("single robot: status=%d sum fz=%.6f (weight %.6f) tau0=%.6f\n", st, fz, w[2], tau[0]);
  if (st != 0 || std::fabs(fz - w[2]) > 1e-2 * w[2]) return 2;
  // batch with raw HIP buffers
  std::vector<double> hq(19 * N), hv(18 * N, 0.0), hw(6 * N, 0.0), ha(18 * N, 0.0), hn(12 * N), hmu(4 * N, 0.6);
  std::vector<int> hmask(N, 0xF);
  for (size_t i = 0; i < N; ++i) {
    for (int c = 0; c < 19; ++c) hq[c * N + i] = q[c];
    hq[2 * N + i] = 0.3 + 0.0001 * i;
    hw[2 * N + i] = w[2];
    for (int c = 0; c < 12; ++c) hn[c * N + i] = nrm[c];
  }
  double *dq, *dv, *dw, *da, *dn, *dmu, *dtau, *df;
  int *dmask, *dst;
  hipMalloc(&dq, hq.size() * 8); hipMalloc(&dv, hv.size() * 8); hipMalloc(&dw, hw.size() * 8); hipMalloc(&da, ha.size() * 8);
  hipMalloc(&dn, hn.size() * 8); hipMalloc(&dmu, hmu.size() * 8); hipMalloc(&dtau, 12 * N * 8); hipMalloc(&df, 12 * N * 8);
  hipMalloc(&dmask, N * 4); hipMalloc(&dst, N * 4);
  hipMemcpy(dq, hq.data(), hq.size() * 8, hipMemcpyHostToDevice); hipMemcpy(dv, hv.data(), hv.size() * 8, hipMemcpyHostToDevice);
  hipMemcpy(dw, hw.data(), hw.size() * 8, hipMemcpyHostToDevice); hipMemcpy(da, ha.data(), ha.size() * 8, hipMemcpyHostToDevice);
  hipMemcpy(dn, hn.data(), hn.size() * 8, hipMemcpyHostToDevice); hipMemcpy(dmu, hmu.data(), hmu.size() * 8, hipMemcpyHostToDevice);
  hipMemcpy(dmask, hmask.data(), N * 4, hipMemcpyHostToDevice);
  wbc_batch_in in = {dq, dv, dw, da, dn, dmu, dmask, nullptr, nullptr};
  wbc_batch_out out = {dtau, df, dst, nullptr, nullptr, nullptr, nullptr, nullptr};
  hipStream_t stream;
  hipStreamCreate(&stream);
  CK(wbc_step_batch(s, N, &in, &out, nullptr, stream));
  hipStreamSynchronize(stream);
  std::vector<double> hf(12 * N);
  std::vector<int> hst(N);
  hipMemcpy(hf.data(), df, hf.size() * 8, hipMemcpyDeviceToHost);
  hipMemcpy(hst.data(), dst, N * 4, hipMemcpyDeviceToHost);
  int bad = 0;
  for (size_t i = 0; i < N; ++i) {
    double z = hf[2 * N + i] + hf[5 * N + i] + hf[8 * N + i] + hf[11 * N + i];
    if (hst[i] != 0 || std::fabs(z - w[2]) > 1e-2 * w[2]) ++bad;
  }
  std::printf("batch %zu: %d bad\n", N, bad);
  {  // a 5-tick rollout of the same batch from C++ (raw buffers): needs M, h, Jc; q, v advance in place
    double *dM, *dh, *dJ;
    int* dit;
    hipMalloc(&dM, 171 * N * 8); hipMalloc(&dh, 18 * N * 8); hipMalloc(&dJ, 216 * N * 8); hipMalloc(&dit, N * 4);
    hipMemset(dtau, 0, 12 * N * 8); hipMemset(df, 0, 12 * N * 8);
    wbc_batch_out ro = {dtau, df, dst, dit, dM, dh, dJ, nullptr};
    CK(wbc_rollout_batch(s, N, 5, &in, &ro, nullptr, nullptr, nullptr, stream));
    hipStreamSynchronize(stream);
    std::vector<double> hq2(19 * N);
    hipMemcpy(hq2.data(), dq, hq2.size() * 8, hipMemcpyDeviceToHost);
    hipMemcpy(hst.data(), dst, N * 4, hipMemcpyDeviceToHost);
    int rbad = 0;
    double dz = 0;
    for (size_t i = 0; i < N; ++i) {
      if (hst[i] != 0) ++rbad;
      const double d = std::fabs(hq2[2 * N + i] - hq[2 * N + i]);   // standing robots asked to hold still: base height barely moves
      if (d > dz) dz = d;
    }
    std::printf("rollout 5 ticks: %d bad, max |dz| = %.2e m\n", rbad, dz);
    if (rbad || !(dz < 1e-3)) return 6;
  }
  wbc_solver_destroy(s);
  wbc_model_free(m);
  if (bad) return 3;
  // the C++ host class: planner call + tick for one robot standing on its goal -> weight-carrying GRFs again
  try {
    wbc::QuadrupedWBC ctl(urdf);
    wbc::BaseState base = {{0, 0, 0.4}, {0, 0, 0, 1}, {0, 0, 0}, {0, 0, 0}};
    wbc::JointState js;
    js.name = ctl.jointNames();
    for (int l = 0; l < 4; ++l) { js.position.insert(js.position.end(), {0.05, 0.75, -1.5}); js.velocity.insert(js.velocity.end(), {0, 0, 0}); }
    // jointNames() order may interleave legs; the nominal posture is the same for every leg only if names map leg-wise
    for (size_t j = 0; j < js.name.size(); ++j) js.position[j] = q[7 + j];
    wbc::ContactState cs;
    for (int k = 0; k < 4; ++k) { cs.stance[k] = true; cs.normal[k][0] = 0; cs.normal[k][1] = 0; cs.normal[k][2] = 1; cs.mu[k] = 0.6; }
    wbc::ComPlan cp = {{0, 0, 0.36}, {0, 0, 0.36}, 0.0, {0, 0, 0, 1}};
    double com[6];
    wbc::Command c0 = ctl.plan(base, js, cp, 0.0, com);      // learn the CoM, then plan to hold it
    for (int k = 0; k < 3; ++k) cp.start[k] = cp.goal[k] = com[k];
    wbc::Command cmd = ctl.plan(base, js, cp, 0.0);
    (void)c0;
    std::vector<double> tq;
    double grf[12];
    int stq = ctl.computeTorques(base, js, cs, cmd, tq, grf);
    double gz = grf[2] + grf[5] + grf[8] + grf[11];
    std::printf("QuadrupedWBC plan+tick: status=%d w_des z=%.4f sum fz=%.4f com=(%.4f %.4f %.4f)\n", stq, cmd.w_des[2], gz, com[0], com[1], com[2]);
    if (stq != 0 || std::fabs(gz - w[2]) > 1e-2 * w[2] || std::fabs(cmd.w_des[2] - w[2]) > 1e-6) return 4;
  } catch (const std::exception& e) {
    std::printf("QuadrupedWBC: %s\n", e.what());
    return 5;
  }
  // the general dense QP through the C-ABI with raw buffers: min 1/2 |x|^2 - x0 - x1  s.t.  x0 + x1 = 1 (equality), x0 >= 0.7  ->  x = (0.7, 0.3)
  {
    const size_t NQ = 3;
    std::vector<double> H(NQ * 4), g(NQ * 2), C(NQ * 4), d(NQ * 2), x(NQ * 2), lam(NQ * 2);
    std::vector<int> qs(NQ), qi(NQ);
    for (size_t k = 0; k < NQ; ++k) {
      H[4 * k] = 1; H[4 * k + 1] = 0; H[4 * k + 2] = 0; H[4 * k + 3] = 1; g[2 * k] = -1; g[2 * k + 1] = -1;
      C[4 * k] = 1; C[4 * k + 1] = 1; C[4 * k + 2] = 1; C[4 * k + 3] = 0; d[2 * k] = 1; d[2 * k + 1] = 0.7 + 0.1 * k;
    }
    double *dH, *dg, *dC, *dd, *dx, *dl;
    int *ds, *di;
    hipMalloc(&dH, H.size() * 8); hipMalloc(&dg, g.size() * 8); hipMalloc(&dC, C.size() * 8); hipMalloc(&dd, d.size() * 8);
    hipMalloc(&dx, x.size() * 8); hipMalloc(&dl, lam.size() * 8); hipMalloc(&ds, NQ * 4); hipMalloc(&di, NQ * 4);
    hipMemcpy(dH, H.data(), H.size() * 8, hipMemcpyHostToDevice); hipMemcpy(dg, g.data(), g.size() * 8, hipMemcpyHostToDevice);
    hipMemcpy(dC, C.data(), C.size() * 8, hipMemcpyHostToDevice); hipMemcpy(dd, d.data(), d.size() * 8, hipMemcpyHostToDevice);
    CK(wbc_qp_dense_batch(WBC_F64, NQ, 2, 2, 1, dH, dg, dC, dd, 50, 1e-10, dx, dl, ds, di, nullptr));
    hipDeviceSynchronize();
    hipMemcpy(x.data(), dx, x.size() * 8, hipMemcpyDeviceToHost); hipMemcpy(qs.data(), ds, NQ * 4, hipMemcpyDeviceToHost);
    hipMemcpy(lam.data(), dl, lam.size() * 8, hipMemcpyDeviceToHost);
    std::printf("wbc_qp_dense_batch: status=%d x=(%.6f %.6f) lambda=(%.4f %.4f)\n", qs[1], x[2], x[3], lam[2], lam[3]);
    for (size_t k = 0; k < NQ; ++k)
      if (qs[k] != 0 || std::fabs(x[2 * k] - (0.7 + 0.1 * k)) > 1e-9 || std::fabs(x[2 * k] + x[2 * k + 1] - 1) > 1e-9 || lam[2 * k + 1] < 0) return 6;
    if (wbc_qp_dense_batch(WBC_F64, NQ, 37, 2, 1, dH, dg, dC, dd, 50, 1e-10, dx, dl, ds, di, nullptr) != WBC_E_INVALID) return 7;
  }
  return 0;
}
