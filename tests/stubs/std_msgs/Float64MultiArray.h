// STUB of std_msgs/Float64MultiArray (field layout only; the MultiArrayLayout member is omitted) -- see tests/stubs/README.md
#pragma once
#include <vector>
namespace std_msgs {
struct Float64MultiArray { std::vector<double> data; };
}
