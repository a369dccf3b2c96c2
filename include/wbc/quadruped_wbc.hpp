// quadruped_wbc.hpp -- header-only C++ host class over the C-ABI (include/wbc_hip.h).
//
// Mirrors the shape of the reference controller's per-tick entry point as far as it is observable:
// a controller object built from the URDF path given on the command line
// (/root/reference/README.md:60) whose tick turns the robot state into joint torques
// (/root/reference/README.md:11).  The reference's class and method names live in an absent
// submodule (/root/reference/.gitmodules:4-6) and are [UNVERIFIED]; rename to taste when binding.
//
// The ROS message adaptors at the bottom are compiled only with -DWBC_WITH_ROS.  ROS is absent from this image: the
// section is compile-checked against field-layout stubs of the three message headers (tests/stubs/, labelled as
// stubs -- they pin nothing about the real controller's topics or types, which are [UNVERIFIED]).
#pragma once
#include <array>
#include <stdexcept>
#include <string>
#include <vector>

#include "../wbc_hip.h"
#ifdef WBC_WITH_ROS
#include <gazebo_msgs/ModelStates.h>
#include <sensor_msgs/JointState.h>
#include <std_msgs/Float64MultiArray.h>
#endif

namespace wbc {

struct Error : std::runtime_error {
  int code;
  Error(int c, const std::string& where)
      : std::runtime_error(where + ": " + wbc_strerror(c) + " (" + wbc_last_error() + ")"), code(c) {}
};
inline void check(int rc, const char* where) { if (rc != WBC_OK) throw Error(rc, where); }

// Plain-data stand-ins with the field layout of the ROS messages a Gazebo quadruped loop uses.
struct BaseState {            // gazebo_msgs/ModelStates entry: pose + twist of the floating base, world frame
  double position[3];
  double orientation_xyzw[4];
  double linear[3];
  double angular[3];
};
struct JointState {           // sensor_msgs/JointState: name[], position[], velocity[]
  std::vector<std::string> name;
  std::vector<double> position, velocity;
};
struct ContactState {         // per-foot stance flag, terrain normal and friction (planner / contact sensors)
  bool stance[4];
  double normal[4][3];
  double mu[4];
};
struct ComPlan {              // input of the CoM planner (wbc_reference_batch's plan row)
  double start[3], goal[3];   // CoM start / goal, world
  double duration;            // s; <= 0 = hold the goal
  double quat_des_xyzw[4];    // desired trunk attitude
};
struct Command {              // what the planner hands to the tick
  double w_des[6];            // desired contact wrench on the base rows
  std::array<double, 18> vdot_des;
};

class QuadrupedWBC {
 public:
  // urdf_path: argv[1] of the reference's `dogbot` executable
  explicit QuadrupedWBC(const std::string& urdf_path, const wbc_params* params = nullptr, int device = 0,
                        const std::vector<std::string>& foot_links = {}) {
    std::vector<const char*> fl;
    for (auto& s : foot_links) fl.push_back(s.c_str());
    check(wbc_model_load_urdf(urdf_path.c_str(), fl.empty() ? nullptr : fl.data(), (int)fl.size(), &model_),
          "wbc_model_load_urdf");
    wbc_params p;
    if (params) p = *params; else wbc_params_default(&p, WBC_F64);
    params_ = p;
    int rc = wbc_solver_create(model_, &p, WBC_F64, device, 1, &solver_);
    if (rc != WBC_OK) { wbc_model_free(model_); model_ = nullptr; throw Error(rc, "wbc_solver_create"); }
    check(wbc_model_dims(model_, nullptr, &nq_, &nv_, &nj_, &nf_), "wbc_model_dims");
    for (int j = 0; j < nj_; ++j) joint_names_.push_back(wbc_model_joint_name(model_, j));
    obs_integ_.assign(nv_, 0.0); obs_r_.assign(nv_, 0.0);
    tau_prev_.assign(nj_, 0.0); f_prev_.assign(3 * nf_, 0.0);
  }
  ~QuadrupedWBC() { if (solver_) wbc_solver_destroy(solver_); if (model_) wbc_model_free(model_); }
  QuadrupedWBC(const QuadrupedWBC&) = delete;
  QuadrupedWBC& operator=(const QuadrupedWBC&) = delete;

  const std::vector<std::string>& jointNames() const { return joint_names_; }

  // One control tick for one robot: returns joint torques in jointNames() order; grf_out (12) optional.
  // Throws on ABI errors; returns the QP status (0 optimal, 1 iteration limit, 2 infeasible).
  int computeTorques(const BaseState& base, const JointState& js, const ContactState& contacts, const Command& cmd,
                     std::vector<double>& tau_out, double* grf_out = nullptr) {
    std::vector<double> q, v;
    packState(base, js, q, v);
    double normals[12], mu[4];
    int mask = 0;
    for (int f = 0; f < nf_; ++f) {
      if (contacts.stance[f]) mask |= 1 << f;
      for (int k = 0; k < 3; ++k) normals[3 * f + k] = contacts.normal[f][k];
      mu[f] = contacts.mu[f];
    }
    tau_out.assign(nj_, 0.0);
    std::vector<double> f(3 * nf_);
    int status = -1;
    const bool obs = params_.observer_order > 0;
    if (obs && !obs_started_) {  // start-up contract of the observer (wbc_observer_state): integ(0) = p(0) = M(q0) v0, r(0) = 0
      check(wbc_observer_init(solver_, q.data(), v.data(), obs_integ_.data(), obs_r_.data()), "wbc_observer_init");
      obs_started_ = true;
    }
    check(wbc_compute_torques(solver_, q.data(), v.data(), cmd.w_des, cmd.vdot_des.data(), normals, mu, mask,
                              obs ? tau_prev_.data() : nullptr, obs ? f_prev_.data() : nullptr,
                              obs ? obs_integ_.data() : nullptr, obs ? obs_r_.data() : nullptr, tau_out.data(), f.data(),
                              &status),
          "wbc_compute_torques");
    tau_prev_ = tau_out; f_prev_ = f;
    if (grf_out) for (int k = 0; k < 3 * nf_; ++k) grf_out[k] = f[k];
    return status;
  }

  // The planner side of the tick (/root/reference/README.md:11 "a motion planner for the trajectory of the robot's
  // center of mass"): turns a CoM plan evaluated `t` seconds after its start into the Command computeTorques tracks.
  // com_out (6, optional) receives the CoM position and velocity of the given state.
  void setReferenceGains(const wbc_ref_params& g) { check(wbc_solver_set_ref_params(solver_, &g), "wbc_solver_set_ref_params"); ref_set_ = true; }
  Command plan(const BaseState& base, const JointState& js, const ComPlan& cp, double t, double* com_out = nullptr) {
    if (!ref_set_) { wbc_ref_params g; wbc_ref_params_default(&g); setReferenceGains(g); }
    std::vector<double> q, v;
    packState(base, js, q, v);
    double row[WBC_PLAN_WORDS];
    for (int k = 0; k < 3; ++k) { row[k] = cp.start[k]; row[3 + k] = cp.goal[k]; }
    row[6] = cp.duration; row[7] = 0.0;
    for (int k = 0; k < 4; ++k) row[8 + k] = cp.quat_des_xyzw[k];
    Command c;
    check(wbc_compute_reference(solver_, q.data(), v.data(), row, t, c.w_des, c.vdot_des.data(), com_out), "wbc_compute_reference");
    return c;
  }

  // Observer state (the only per-robot state carried across ticks): snapshot / restore / initialise.
  void setObserverState(const std::vector<double>& integ, const std::vector<double>& r) { obs_integ_ = integ; obs_r_ = r; obs_started_ = true; }
  void resetObserver() { obs_started_ = false; }   // the next computeTorques re-seeds integ = M v, r = 0
  const std::vector<double>& disturbanceEstimate() const { return obs_r_; }

 private:
  void packState(const BaseState& base, const JointState& js, std::vector<double>& q, std::vector<double>& v) const {
    q.assign(nq_, 0.0); v.assign(nv_, 0.0);
    for (int k = 0; k < 3; ++k) { q[k] = base.position[k]; v[k] = base.linear[k]; v[3 + k] = base.angular[k]; }
    for (int k = 0; k < 4; ++k) q[3 + k] = base.orientation_xyzw[k];
    // JointState carries names: map by name, as ROS controllers do
    for (int j = 0; j < nj_; ++j) {
      bool found = false;
      for (size_t i = 0; i < js.name.size(); ++i)
        if (js.name[i] == joint_names_[j]) { q[7 + j] = js.position[i]; v[6 + j] = js.velocity[i]; found = true; break; }
      if (!found) throw std::invalid_argument("JointState lacks joint " + joint_names_[j]);
    }
  }
  bool ref_set_ = false;
  wbc_model* model_ = nullptr;
  wbc_solver* solver_ = nullptr;
  wbc_params params_;
  int nq_ = 0, nv_ = 0, nj_ = 0, nf_ = 0;
  bool obs_started_ = false;
  std::vector<std::string> joint_names_;
  std::vector<double> obs_integ_, obs_r_, tau_prev_, f_prev_;
};

#ifdef WBC_WITH_ROS
// Compile-checked against stubs only (ROS absent from the build image).  Adaptors from the standard message types of a
// Gazebo + ros_control loop (/root/reference/README.md:38 names ros-control / ros-controllers as dependencies):
//   BaseState  <- gazebo_msgs::ModelStates (pose[i], twist[i] of the robot model)
//   JointState <- sensor_msgs::JointState
//   torques    -> std_msgs::Float64MultiArray for a JointGroupEffortController (ros_controllers), jointNames() order
// index of the robot in a ModelStates message; -1 when absent
inline int findModel(const gazebo_msgs::ModelStates& ms, const std::string& model_name) {
  for (size_t i = 0; i < ms.name.size(); ++i) if (ms.name[i] == model_name) return (int)i;
  return -1;
}
inline BaseState fromRos(const gazebo_msgs::ModelStates& ms, size_t i) {
  BaseState b;
  b.position[0] = ms.pose[i].position.x; b.position[1] = ms.pose[i].position.y; b.position[2] = ms.pose[i].position.z;
  b.orientation_xyzw[0] = ms.pose[i].orientation.x; b.orientation_xyzw[1] = ms.pose[i].orientation.y;
  b.orientation_xyzw[2] = ms.pose[i].orientation.z; b.orientation_xyzw[3] = ms.pose[i].orientation.w;
  b.linear[0] = ms.twist[i].linear.x; b.linear[1] = ms.twist[i].linear.y; b.linear[2] = ms.twist[i].linear.z;
  b.angular[0] = ms.twist[i].angular.x; b.angular[1] = ms.twist[i].angular.y; b.angular[2] = ms.twist[i].angular.z;
  return b;
}
inline JointState fromRos(const sensor_msgs::JointState& m) { return JointState{m.name, m.position, m.velocity}; }
inline std_msgs::Float64MultiArray toRos(const std::vector<double>& tau) {
  std_msgs::Float64MultiArray cmd;
  cmd.data = tau;
  return cmd;
}
#endif

}  // namespace wbc
