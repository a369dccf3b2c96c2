// CPU sanitizer check of the host-side URDF reader (no HIP): g++ -fsanitize=address,undefined tools/asan_urdf.cpp wbc_quadruped_dob_amd/csrc/urdf_reader.cpp
#include <cstdio>
#include <fstream>
#include <string>
#include <vector>
#include "../wbc_quadruped_dob_amd/csrc/model.hpp"
int main(int argc, char** argv) {
  wbc::FlatModel m;
  std::string err;
  int rc = wbc::load_urdf(argv[1], {}, m, err);
  std::printf("good file: rc=%d nb=%d nf=%d\n", rc, m.nb, m.nf());
  int lb[4][3];
  std::printf("topology rc=%d\n", wbc::quadruped_topology(m, lb, err));
  // hostile inputs: truncations of the real file at every 97th byte, and junk
  std::ifstream f(argv[1]);
  std::string text((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
  int bad = 0;
  for (size_t cut = 0; cut < text.size(); cut += 97) {
    std::ofstream o("/tmp/_cut.urdf");
    o << text.substr(0, cut);
    o.close();
    wbc::FlatModel t;
    if (wbc::load_urdf("/tmp/_cut.urdf", {}, t, err) != 0) ++bad;
  }
  const char* junk[] = {"", "<", "<robot", "<robot><link name='a'><inertial><mass value='x'/></inertial></link></robot>",
                        "<robot><joint name='j' type='revolute'><parent link='a'/><child link='a'/></joint><link name='a'/></robot>",
                        "<robot><link name='a'/><link name='b'/><joint name='j' type='fixed'><parent link='a'/><child link='b'/></joint>"
                        "<joint name='k' type='fixed'><parent link='b'/><child link='a'/></joint></robot>"};
  for (const char* j : junk) {
    std::ofstream o("/tmp/_cut.urdf");
    o << j;
    o.close();
    wbc::FlatModel t;
    int r = wbc::load_urdf("/tmp/_cut.urdf", {}, t, err);
    std::printf("junk -> rc=%d (%s)\n", r, err.c_str());
  }
  std::printf("truncations rejected: %d\n", bad);
  return 0;
}
