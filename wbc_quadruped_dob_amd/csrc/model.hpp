// Host-side flattened robot model (product code; independent of oracle/).
// Layout documented in include/wbc_hip.h and DESIGN.md section 2.
#pragma once
#include <string>
#include <vector>

namespace wbc {

struct FlatModel {
  int nb = 0;                  // bodies incl. floating base (body 0)
  std::vector<int> parent;     // [nb]
  std::vector<double> Rt;      // [nb*9] child(q=0) -> parent rotation, row-major
  std::vector<double> rt;      // [nb*3] joint origin in parent coordinates
  std::vector<double> axis;    // [nb*3] unit joint axis in child frame
  std::vector<double> mass;    // [nb]
  std::vector<double> com;     // [nb*3]
  std::vector<double> Ic;      // [nb*6] xx,xy,xz,yy,yz,zz about COM, link axes
  std::vector<int> foot_body;  // [nf]
  std::vector<double> foot_off;  // [nf*3]
  double gravity[3] = {0.0, 0.0, -9.81};
  std::vector<std::string> joint_names;  // [nb-1]
  std::vector<std::string> foot_links;   // [nf]
  std::vector<std::string> body_names;   // [nb]
  int nf() const { return (int)foot_body.size(); }
  int nj() const { return nb - 1; }
  int nv() const { return 6 + nb - 1; }
  int nq() const { return 7 + nb - 1; }
};

// Reads the URDF subset {link/inertial, joint revolute|continuous|fixed, origin, axis}.
// Fixed-joint children are lumped into their parent body.  Returns a wbc_status code and
// fills `err` with a human-readable reason on failure.
int load_urdf(const std::string& path, const std::vector<std::string>& foot_links, FlatModel& out, std::string& err);

// Quadruped kernel topology: 4 legs, each a serial chain of 3 revolute joints hanging off the
// base, one foot on each distal body.  leg_body[l][k] = body index of joint k of leg l, legs in
// order of their feet.  Returns WBC_OK or WBC_E_TOPOLOGY.
int quadruped_topology(const FlatModel& m, int leg_body[4][3], std::string& err);

}  // namespace wbc
