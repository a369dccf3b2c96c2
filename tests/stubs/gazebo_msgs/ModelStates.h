// STUB of gazebo_msgs/ModelStates (field layout only) -- see tests/stubs/README.md
#pragma once
#include <string>
#include <vector>
#include <geometry_msgs/Pose.h>
#include <geometry_msgs/Twist.h>
namespace gazebo_msgs {
struct ModelStates {
  std::vector<std::string> name;
  std::vector<geometry_msgs::Pose> pose;
  std::vector<geometry_msgs::Twist> twist;
};
}
