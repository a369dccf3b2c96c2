// rollout_kernel launches: a whole horizon of dependent ticks incl. forward dynamics as one launch (fused_tick.hip.hpp).
// The planner-in-the-loop instantiations (-DWBC_ROLLOUT_TRACK=1) compile as their own unit.
#include "k_common.hip.hpp"
#include "fused_tick.hip.hpp"

namespace wbc {

#ifndef WBC_ROLLOUT_TRACK
#define WBC_ROLLOUT_TRACK 0
#endif
#define WBC_ROLLOUT_ARGS const LaunchCtx& L, bool observer, int spw, const DevModel<Scalar>* model, const DevParams<Scalar>& prm,            \
                         const SweepArgs<Scalar>& a, const QpArgs<Scalar>& qa, const QpJidx& jmap, const IntegrateArgs<Scalar>& ia, int horizon, \
                         const DevRefParams<Scalar>* G, const RefArgs<Scalar>& ra, bool warm
hipError_t rollout_plain(WBC_ROLLOUT_ARGS);
hipError_t rollout_track(WBC_ROLLOUT_ARGS);

// (4-state workgroups are four wavefronts since round 5: WBC_RO_MERGE, fused_tick.hip.hpp)
#define WBC_ROLLOUT_THREADS(OB_, SPW_) rollout_threads(OB_, SPW_)
#define WBC_ROLLOUT(OB_, SPW_) \
  do { if (warm) WBC_KLAUNCH(L, (rollout_kernel<T, OB_, (WBC_ROLLOUT_TRACK != 0), SPW_, true>), grid, dim3(WBC_ROLLOUT_THREADS(OB_, SPW_)), model, prm, a, qa, jmap, ia, horizon, G, ra); \
       else WBC_KLAUNCH(L, (rollout_kernel<T, OB_, (WBC_ROLLOUT_TRACK != 0), SPW_, false>), grid, dim3(WBC_ROLLOUT_THREADS(OB_, SPW_)), model, prm, a, qa, jmap, ia, horizon, G, ra); } while (0)

#if WBC_ROLLOUT_TRACK
hipError_t rollout_track(WBC_ROLLOUT_ARGS) {
#else
hipError_t rollout_plain(WBC_ROLLOUT_ARGS) {
#endif
  using T = Scalar;
  const dim3 grid((unsigned)((a.N + spw - 1) / spw));
  if (spw == 4) { if (observer) WBC_ROLLOUT(true, 4); else WBC_ROLLOUT(false, 4); }
  else { if (observer) WBC_ROLLOUT(true, 16); else WBC_ROLLOUT(false, 16); }
  return hipGetLastError();
}

#if !WBC_ROLLOUT_TRACK
template <>
hipError_t k_rollout<Scalar>(const LaunchCtx& L, bool observer, bool track, int spw, const DevModel<Scalar>* model, const DevParams<Scalar>& prm,
                             const SweepArgs<Scalar>& a, const QpArgs<Scalar>& qa, const QpJidx& jmap, const IntegrateArgs<Scalar>& ia, int horizon,
                             const DevRefParams<Scalar>* G, const RefArgs<Scalar>& ra, bool warm) {
  return track ? rollout_track(L, observer, spw, model, prm, a, qa, jmap, ia, horizon, G, ra, warm)
               : rollout_plain(L, observer, spw, model, prm, a, qa, jmap, ia, horizon, G, ra, warm);
}
#endif

}  // namespace wbc
