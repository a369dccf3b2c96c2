// STUB of sensor_msgs/JointState (field layout only) -- see tests/stubs/README.md
#pragma once
#include <string>
#include <vector>
namespace sensor_msgs {
struct JointState {
  std::vector<std::string> name;
  std::vector<double> position, velocity, effort;
};
}
