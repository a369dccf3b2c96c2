// STUB of geometry_msgs/Pose (field layout only) -- see tests/stubs/README.md
#pragma once
namespace geometry_msgs {
struct Point { double x = 0, y = 0, z = 0; };
struct Quaternion { double x = 0, y = 0, z = 0, w = 1; };
struct Pose { Point position; Quaternion orientation; };
}
