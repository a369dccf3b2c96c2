// qp_general_kernel: a strictly convex dense QP of RUN-TIME size, one per WAVEFRONT, factors and working set in LDS.
//
//     min 1/2 x^T H x + g^T x     s.t.   C_i x  = d_i  (i <  meq),     C_i x >= d_i  (meq <= i < m),      n <= 36,  m <= 64
//
// Why it exists: the GRF QP of this controller (units a7 / a8) is the structured 12-variable problem the kernels of
// qp_struct16.hip.hpp / qp_lane.hip.hpp are written around, and that shape is a GUESS about the reference: README.md:11 says only
// "optimization problem based on the modulation of ground reaction forces", the controller's source is an absent submodule
// (.gitmodules:4-6).  A formulation with other variables (accelerations, slacks, joint-torque rows) does not fit those kernels at
// all.  This one takes any (H, g, C, d) -- the survey's "parametric in nvar <= 36, ncon <= 48" (64 rows here: a constraint per lane), and the north_star's "one QP per
// wavefront with active-set iterations held in LDS" literally.  It is the general path, not the fast one (the 12-variable GRF QP
// through it: see docs/DESIGN_R04.md section 4.3c for the measured factor).
//
// Method: Goldfarb-Idnani dual active set, as oracle/qp_general.hpp (same steps, same tests, same status codes, so iteration
// counts agree) with the wavefront's 64 lanes as the vector unit:
//   lane i  <-> variable i (x_i, row i of J), constraint i (its slack, d_i, active flag) and active-set SLOT i (constraint id, u, sign)
//   LDS     <-> J = L^-T Q (n x n), R (n x n, upper), C (m x n), the vectors every lane reads (x, np, dd, z, rotation coefficients)
//   matrices are stored with an ODD leading dimension: a lane-per-row access (stride ld) and a lane-per-column access (stride 1)
//   are both bank-conflict-free.
// Two places depart from the oracle's operation ORDER (not from its mathematics; results agree to rounding):
//   * r = R^-1 d1 by column-oriented back substitution (one broadcast per column instead of a dot product per row);
//   * the Givens chain that rotates d2 into its first component: its coefficients are c_j = d_{j-1} / s_{j-1}, s_j / s_{j-1} with the
//     suffix norms s_j = |d_{j..n-1}| (s_{n-1} = d_{n-1}, signed), so they are all known up front (one reverse scan) and every lane then applies the whole chain to
//     ITS row of J as a private recurrence -- no cross-lane dependency, LDS traffic pipelined -- instead of n - q - 1 dependent
//     wavefront-wide steps.
// One wavefront works alone on its QP: every branch below is wavefront-uniform, the only synchronisation is the program order of
// one wavefront's LDS operations (made explicit for the compiler by wsync()).
#pragma once
#include <hip/hip_runtime.h>
#include <type_traits>
#include "qp_general_args.hpp"
#include "qp_group16.hip.hpp"   // dppx, gsum, KeyT (DPP helpers of the 16-lane solver)

namespace wbc {

template <class T> struct QpgLim;
template <> struct QpgLim<double> { static constexpr double eps = 2.220446049250313e-16, big = 1.0e300; };
template <> struct QpgLim<float> { static constexpr float eps = 1.1920929e-07f, big = 1.0e30f; };

#define QPG_WSYNC() do { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); } while (0)

// Cross-lane plumbing of one wavefront, DPP and v_readlane only (a ds_bpermute round trip costs ~10x a DPP move, and the loop below
// is a chain of such steps): reductions go 4 DPP steps inside each 16-lane row, then the four row results are combined through
// v_readlane (wavefront-uniform values).
__device__ __forceinline__ double qpg_rl(double v, int lane) {   // `lane` must be wavefront-uniform
  const long long b = __double_as_longlong(v);
  const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(b & 0xFFFFFFFFll), lane);
  const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)((unsigned long long)b >> 32), lane);
  return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
__device__ __forceinline__ float qpg_rl(float v, int lane) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane)); }
__device__ __forceinline__ int qpg_rl(int v, int lane) { return __builtin_amdgcn_readlane(v, lane); }
// value of a lane whose index every lane agrees on but the compiler cannot know to be uniform
template <class T> __device__ __forceinline__ T qpg_rl_dyn(T v, int lane) { return qpg_rl(v, __builtin_amdgcn_readfirstlane(lane)); }
template <class T> __device__ __forceinline__ T qpg_wave_sum(T v) {
  v = gsum(v);
  return (qpg_rl(v, 0) + qpg_rl(v, 16)) + (qpg_rl(v, 32) + qpg_rl(v, 48));
}
// minimum and (one of) the lane(s) that hold it, the lowest on an exact tie; BIG = "no candidate".  Selection by the packed key of
// qp_group16.hip.hpp (lane id in the low mantissa bits), the EXACT value of the winner is then read from its lane.
template <class T> __device__ __forceinline__ void qpg_wave_argmin(T& v, int& idx) {
  using K = KeyT<T>;
  T k = K::pack(v, idx);
  k = K::mn(k, dppx<0xB1>(k)); k = K::mn(k, dppx<0x4E>(k)); k = K::mn(k, dppx<0x141>(k)); k = K::mn(k, dppx<0x140>(k));
  const T k0 = qpg_rl(k, 0), k1 = qpg_rl(k, 16), k2 = qpg_rl(k, 32), k3 = qpg_rl(k, 48);
  const T ka = k0 < k1 ? k0 : k1, kb = k2 < k3 ? k2 : k3;
  const T kk = ka < kb ? ka : kb;
  idx = K::id(kk);
  v = qpg_rl_dyn(v, idx);
}
// lane i <- lane i-1 / lane i+1 over the whole wavefront (wave_shr:1 / wave_shl:1; the end lane keeps its own value)
template <class T> __device__ __forceinline__ T qpg_from_below(T v) { return dppx<0x138>(v); }
template <class T> __device__ __forceinline__ T qpg_from_above(T v) { return dppx<0x130>(v); }
// inclusive SUFFIX sum over the lanes: out_j = sum_{k >= j} v_k  (row_shl 1, 2, 4, 8 with zero fill inside the rows, then the totals of the
// rows above -- lane 0 of each row holds its row's total)
template <class T> __device__ __forceinline__ T qpg_suffix_sum(T v, int lane) {
  auto shl = [](T x, auto ctl) __attribute__((always_inline)) {
    constexpr int C = decltype(ctl)::value;
    if constexpr (std::is_same<T, double>::value) {
      const long long xi = __double_as_longlong(x);
      return __longlong_as_double(__builtin_amdgcn_update_dpp(0ll, xi, C, 0xF, 0xF, true));
    } else {
      return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), C, 0xF, 0xF, true));
    }
  };
  v += shl(v, std::integral_constant<int, 0x101>()); v += shl(v, std::integral_constant<int, 0x102>());
  v += shl(v, std::integral_constant<int, 0x104>()); v += shl(v, std::integral_constant<int, 0x108>());
  const T t1 = qpg_rl(v, 16), t2 = qpg_rl(v, 32), t3 = qpg_rl(v, 48);
  const int row = lane >> 4;
  return v + (row == 0 ? (t1 + t2) + t3 : (row == 1 ? t2 + t3 : (row == 2 ? t3 : (T)0)));
}

// sum_{k0 <= k < k1} a[k * sa] * b[k * sb] with four independent partial sums: the LDS reads of a trip are in flight together and the
// FMA chain is a quarter as long (a lone accumulator pays ~80 cycles of LDS latency plus a dependent FMA per term)
template <class T> __device__ __forceinline__ T qpg_dot(const T* a, int sa, const T* b, int sb, int k0, int k1) {
  T s0 = 0, s1 = 0, s2 = 0, s3 = 0;
  int k = k0;
  for (; k + 4 <= k1; k += 4) {
    s0 += a[k * sa] * b[k * sb]; s1 += a[(k + 1) * sa] * b[(k + 1) * sb];
    s2 += a[(k + 2) * sa] * b[(k + 2) * sb]; s3 += a[(k + 3) * sa] * b[(k + 3) * sb];
  }
  for (; k < k1; ++k) s0 += a[k * sa] * b[k * sb];
  return (s0 + s1) + (s2 + s3);
}

template <class T>
__global__ __launch_bounds__(256) void qp_general_kernel(QpGeneralArgs<T> a, int lds_per_qp) {
  extern __shared__ __align__(16) unsigned char qpg_lds_raw[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wpb = blockDim.x >> 6;
  const size_t qp = (size_t)blockIdx.x * wpb + wave;
  if (qp >= a.N) return;   // (whole wavefronts leave: no workgroup barrier anywhere below)
  const int n = a.n, m = a.m, meq = a.meq, ld = qpg_ld(n);
  T* const S = (T*)qpg_lds_raw + (size_t)wave * lds_per_qp;
  T* const J = S;                      // n x ld
  T* const R = J + n * ld;             // n x ld   (holds H, then L, until the first constraint is added)
  T* const Cm = R + n * ld;            // m x ld
  T* const xs = Cm + m * ld;           // n+1 each:
  T* const nps = xs + (n + 1);
  T* const dds = nps + (n + 1);
  T* const zs = dds + (n + 1);
  T* const ccs = zs + (n + 1);
  T* const sss = ccs + (n + 1);
  T* const rdg = sss + (n + 1);        // reciprocals of the diagonal of R
  T* const lam = rdg + (n + 1);        // m+1
  const T INF = QpgLim<T>::big, eps = QpgLim<T>::eps;

  // ---- inputs (coalesced: consecutive lanes, consecutive elements of the problem's own block)
  const T* Hq = a.H + qp * (size_t)n * n;
  for (int e = lane; e < n * n; e += 64) { const int i = e / n, j = e - i * n; R[i * ld + j] = Hq[e]; }
  const T* Cq = a.C + qp * (size_t)m * n;
  for (int e = lane; e < m * n; e += 64) { const int i = e / n, j = e - i * n; Cm[i * ld + j] = Cq[e]; }
  const T g_i = lane < n ? a.g[qp * n + lane] : (T)0;
  const T d_i = lane < m ? a.d[qp * m + lane] : (T)0;
  if (lane <= m) lam[lane] = 0;
  int status = 0, iter = 0;
  QPG_WSYNC();

  // ---- Cholesky H = L L^T in place (lower triangle of the R area), column by column; lane i owns row i
  bool notpd = false;
  for (int j = 0; j < n; ++j) {
    T s = 0;
    if (lane >= j && lane < n) {
      s = R[lane * ld + j];
      s -= qpg_dot(R + lane * ld, 1, R + j * ld, 1, 0, j);
    }
    const T sjj = qpg_rl_dyn(s, j);
    if (!(sjj > 0)) { notpd = true; break; }
    const T ljj = sqrt(sjj);
    if (lane >= j && lane < n) R[lane * ld + j] = (lane == j) ? ljj : s / ljj;
    QPG_WSYNC();
  }
  if (notpd) {   // not positive definite: nothing to solve
    if (lane < n) a.x[qp * n + lane] = 0;
    if (a.lambda && lane < m) a.lambda[qp * m + lane] = 0;
    if (lane == 0) { a.status[qp] = 3; if (a.iters) a.iters[qp] = 0; }
    return;
  }
  // ---- J = L^-T (upper triangular): lane c solves L^T J[:, c] = e_c from the bottom up
  if (lane < n) {
    const int c = lane;
    for (int i = n - 1; i >= 0; --i) {
      T s = (i == c) ? (T)1 : (T)0;
      if (i <= c) { for (int k = i + 1; k <= c; ++k) s -= R[k * ld + i] * J[k * ld + c]; s /= R[i * ld + i]; }
      else s = 0;
      J[i * ld + c] = s;
    }
  }
  QPG_WSYNC();
  // ---- unconstrained minimum x = -J J^T g
  T x_i = 0;
  {
    if (lane < n) nps[lane] = g_i;
    QPG_WSYNC();
    T t = 0;
    if (lane < n) t = qpg_dot(J + lane, ld, nps, 1, 0, n);
    if (lane < n) zs[lane] = t;
    QPG_WSYNC();
    if (lane < n) { x_i = -qpg_dot(J + lane * ld, 1, zs, 1, 0, n); xs[lane] = x_i; }
    QPG_WSYNC();
  }

  // ---- dual active-set iterations
  int iq = 0, neq_in = 0, next_eq = 0;
  bool active_i = false;        // constraint `lane` is in the active set
  int A_s = -1;                 // slot `lane`: constraint id, multiplier, sign of its normal
  T u_s = 0, sg_s = 1;
  T Rnorm = 1;
  bool finished = false;
  while (!finished) {
    // step 1: the next equality row, else the most violated inactive inequality
    T s_i = 0;
    if (lane < m) s_i = qpg_dot(Cm + lane * ld, 1, xs, 1, 0, n) - d_i;
    int ip; T sip, sign = 1; bool is_eq = false;
    if (next_eq < meq) {
      const T s = qpg_rl_dyn(s_i, next_eq);
      ip = next_eq++; is_eq = true; sign = s > 0 ? (T)-1 : (T)1; sip = -fabs(s);
    } else {
      T v = (lane >= meq && lane < m && !active_i && s_i < -a.tol) ? s_i : INF;
      int id = lane;
      qpg_wave_argmin(v, id);
      if (!(v < INF)) break;
      ip = id; sip = v;
    }
    const T np_i = lane < n ? sign * Cm[ip * ld + lane] : (T)0;
    if (lane < n) nps[lane] = np_i;
    if (lane == iq) { A_s = ip; u_s = 0; sg_s = sign; }
    QPG_WSYNC();
    // step 2
    for (;;) {
      if (++iter > a.max_iter) { status = 1; finished = true; break; }
      T dd_i = 0;
      if (lane < n) dd_i = qpg_dot(J + lane, ld, nps, 1, 0, n);
      if (lane < n) dds[lane] = dd_i;
      QPG_WSYNC();
      T z_i = 0;
      if (lane < n) z_i = qpg_dot(J + lane * ld, 1, dds, 1, iq, n);
      // r = R^-1 d1 (column-oriented): lane i < iq
      T r_s = 0;
      {
        T s = lane < iq ? dd_i : (T)0;
        for (int j = iq - 1; j >= 0; --j) {
          const T rj = qpg_rl_dyn(s, j) * rdg[j];
          if (lane == j) r_s = rj;
          if (lane < j) s -= R[lane * ld + j] * rj;
        }
      }
      // ratio test over the inequality slots
      T t1 = (lane >= neq_in && lane < iq && r_s > 0) ? u_s / r_s : INF;
      int lslot = lane;
      qpg_wave_argmin(t1, lslot);
      const T dn2 = qpg_wave_sum((lane >= iq && lane < n) ? dd_i * dd_i : (T)0);
      const T znp = qpg_wave_sum(z_i * np_i);
      T t2 = INF;
      if (dn2 > (eps * Rnorm) * (eps * Rnorm) && znp > 0) t2 = -sip / znp;
      const bool no1 = !(t1 < INF), no2 = !(t2 < INF);
      bool do_drop = false;
      if (no1 && no2) {
        if (is_eq && -sip <= a.tol) {   // dependent equality row that already holds: leave it out of the factors
          if (lane == iq) { A_s = -1; u_s = 0; sg_s = 1; }
          if (lane == ip) active_i = true;
          break;
        }
        status = 2; finished = true; break;
      }
      if (no2) {   // dual step only
        if (lane < iq) u_s -= t1 * r_s;
        if (lane == iq) u_s += t1;
        do_drop = true;
      } else {
        const bool full = !(t1 < t2);
        const T t = full ? t2 : t1;
        if (lane < n) { x_i += t * z_i; xs[lane] = x_i; }
        if (lane < iq) u_s -= t * r_s;
        if (lane == iq) u_s += t;
        if (full) {
          // rotate d2 = dd[iq .. n) into its first component; same rotations on the columns of J (see the header)
          const T sq = (lane >= iq && lane < n) ? dd_i * dd_i : (T)0;
          const T suf = qpg_suffix_sum(sq, lane);   // suf_j = sum_{k >= j} dd_k^2
          const T sig = sqrt(suf);
          const T sig_lo = qpg_from_below(sig), dd_lo = qpg_from_below(dd_i);   // sigma_{j-1}, dd_{j-1}
          if (lane > iq && lane < n) {
            const bool idn = !(sig_lo > 0);
            ccs[lane] = idn ? (T)1 : dd_lo / sig_lo;
            sss[lane] = idn ? (T)0 : (lane == n - 1 ? dd_i : sig) / sig_lo;   // (the chain starts from the LAST component itself, sign and all)
          }
          // the new R diagonal: the norm of d2 -- except when d2 has ONE component (iq = n - 1): no rotation happens, the component
          // keeps its sign (R[iq][iq] must stay J[:, iq] . n+)
          const T dq = qpg_rl_dyn((iq == n - 1) ? dd_i : sig, iq);
          QPG_WSYNC();
          if (lane < n) {
            T t_hi = J[lane * ld + (n - 1)];
            for (int j = n - 1; j > iq; --j) {
              const T cc = ccs[j], ss = sss[j];
              const T a1 = J[lane * ld + j - 1];
              J[lane * ld + j] = -ss * a1 + cc * t_hi;
              t_hi = cc * a1 + ss * t_hi;
            }
            J[lane * ld + iq] = t_hi;
          }
          if (lane < iq) R[lane * ld + iq] = dd_i;
          if (lane == iq) { R[iq * ld + iq] = dq; rdg[iq] = (T)1 / dq; }
          Rnorm = fmax(Rnorm, fabs(dq));
          if (lane == ip) active_i = true;
          if (is_eq) ++neq_in;
          ++iq;
          QPG_WSYNC();
          break;
        }
        do_drop = true;
      }
      if (do_drop) {   // slot lslot leaves the active set; the candidate moves down with the slots behind it
        const int qq = lslot;
        const int lcon = qpg_rl_dyn(A_s, qq);
        if (lane == lcon) active_i = false;
        {
          const int A_n = qpg_from_above(A_s); const T u_n = qpg_from_above(u_s), sg_n = qpg_from_above(sg_s);
          if (lane >= qq && lane < iq) { A_s = A_n; u_s = u_n; sg_s = sg_n; }
          if (lane == iq) { A_s = -1; u_s = 0; sg_s = 1; }
        }
        if (lane < n) for (int c = qq; c < iq - 1; ++c) R[lane * ld + c] = R[lane * ld + c + 1];   // my row: columns shift left
        --iq;
        QPG_WSYNC();
        for (int j = qq; j < iq; ++j) {   // re-triangularise the Hessenberg part
          T cc = R[j * ld + j], ss = R[(j + 1) * ld + j];
          const T h = sqrt(cc * cc + ss * ss);
          if (h == 0) continue;
          cc /= h; ss /= h;
          QPG_WSYNC();   // every lane has read the pair before anybody overwrites it
          if (lane == j) { R[(j + 1) * ld + j] = 0; R[j * ld + j] = h; rdg[j] = (T)1 / h; }
          if (lane > j && lane < iq) {
            const T t1_ = R[j * ld + lane], t2_ = R[(j + 1) * ld + lane];
            R[j * ld + lane] = cc * t1_ + ss * t2_;
            R[(j + 1) * ld + lane] = -ss * t1_ + cc * t2_;
          }
          if (lane < n) {
            const T t1_ = J[lane * ld + j], t2_ = J[lane * ld + j + 1];
            J[lane * ld + j] = cc * t1_ + ss * t2_;
            J[lane * ld + j + 1] = -ss * t1_ + cc * t2_;
          }
          QPG_WSYNC();
        }
        if (!no2) {   // it was a partial step: the candidate's slack at the new point
          T s = 0;
          if (lane < n) s = Cm[ip * ld + lane] * x_i;
          s = qpg_wave_sum(s) - qpg_rl_dyn(d_i, ip);
          sip = sign * s;
        }
      }
    }
  }

  // ---- outputs
  if (lane < iq && A_s >= 0) lam[A_s] = sg_s * u_s;
  QPG_WSYNC();
  if (lane < n) a.x[qp * n + lane] = x_i;
  if (a.lambda && lane < m) a.lambda[qp * m + lane] = lam[lane];
  if (lane == 0) { a.status[qp] = status; if (a.iters) a.iters[qp] = iter; }
}

}  // namespace wbc
