// Compile-and-run check of the WBC_WITH_ROS adaptors against the message STUBS of this directory (no GPU, no ROS).
#define WBC_WITH_ROS 1
#include <wbc/quadruped_wbc.hpp>
#include <cstdio>
int main() {
  gazebo_msgs::ModelStates ms;
  ms.name = {"ground_plane", "dogbot"};
  ms.pose.resize(2); ms.twist.resize(2);
  ms.pose[1].position.z = 0.4; ms.pose[1].orientation.w = 1.0;
  ms.twist[1].linear.x = 0.25; ms.twist[1].angular.z = -0.5;
  const int i = wbc::findModel(ms, "dogbot");
  if (i != 1 || wbc::findModel(ms, "absent") != -1) return 1;
  const wbc::BaseState b = wbc::fromRos(ms, (size_t)i);
  if (b.position[2] != 0.4 || b.orientation_xyzw[3] != 1.0 || b.linear[0] != 0.25 || b.angular[2] != -0.5) return 2;
  sensor_msgs::JointState js;
  js.name = {"a", "b"}; js.position = {0.1, 0.2}; js.velocity = {1.0, 2.0};
  const wbc::JointState j = wbc::fromRos(js);
  if (j.name.size() != 2 || j.position[1] != 0.2 || j.velocity[0] != 1.0) return 3;
  const std_msgs::Float64MultiArray cmd = wbc::toRos({1.5, -2.5});
  if (cmd.data.size() != 2 || cmd.data[1] != -2.5) return 4;
  std::puts("ros adaptors ok (stubs)");
  return 0;
}
