// Dependency-free URDF-subset reader (no XML/URDF library exists in this image).
// Stands for the reference controller's model ingestion from argv[1]
// (/root/reference/README.md:60); the reference's loader itself is in an absent submodule.
#include "model.hpp"
#include "../../include/wbc_hip.h"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <map>
#include <memory>
#include <sstream>

namespace wbc {
namespace {

struct Elem {
  std::string name;
  std::map<std::string, std::string> attr;
  std::vector<std::unique_ptr<Elem>> kids;
  const Elem* child(const char* n) const {
    for (auto& k : kids) if (k->name == n) return k.get();
    return nullptr;
  }
  std::string get(const char* k, const char* dflt) const {
    auto it = attr.find(k);
    return it == attr.end() ? std::string(dflt) : it->second;
  }
};

// Minimal XML tokenizer: elements, attributes, comments, <? ?> and <! > declarations; text ignored.
class Xml {
 public:
  explicit Xml(const std::string& s) : s_(s) {}
  bool parse(Elem& root, std::string& err) {
    std::vector<Elem*> stack;
    bool have_root = false;
    while (true) {
      size_t lt = s_.find('<', i_);
      if (lt == std::string::npos) break;
      i_ = lt + 1;
      if (starts("!--")) {
        size_t e = s_.find("-->", i_);
        if (e == std::string::npos) { err = "unterminated comment"; return false; }
        i_ = e + 3;
        continue;
      }
      if (starts("?") || starts("!")) {
        size_t e = s_.find('>', i_);
        if (e == std::string::npos) { err = "unterminated declaration"; return false; }
        i_ = e + 1;
        continue;
      }
      if (starts("/")) {
        ++i_;
        std::string n = ident();
        skip_ws();
        if (i_ >= s_.size() || s_[i_] != '>') { err = "malformed closing tag </" + n; return false; }
        ++i_;
        if (stack.empty() || stack.back()->name != n) { err = "mismatched closing tag </" + n + ">"; return false; }
        stack.pop_back();
        continue;
      }
      std::string n = ident();
      if (n.empty()) { err = "empty tag name"; return false; }
      Elem* e;
      if (stack.empty()) {
        if (have_root) { err = "multiple root elements"; return false; }
        have_root = true;
        root.name = n;
        e = &root;
      } else {
        stack.back()->kids.emplace_back(new Elem);
        e = stack.back()->kids.back().get();
        e->name = n;
      }
      // attributes
      while (true) {
        skip_ws();
        if (i_ >= s_.size()) { err = "unterminated tag <" + n; return false; }
        if (s_[i_] == '/') {
          if (i_ + 1 >= s_.size() || s_[i_ + 1] != '>') { err = "malformed tag <" + n; return false; }
          i_ += 2;
          break;  // self-closing
        }
        if (s_[i_] == '>') { ++i_; stack.push_back(e); break; }
        std::string k = ident();
        skip_ws();
        if (k.empty() || i_ >= s_.size() || s_[i_] != '=') { err = "malformed attribute in <" + n + ">"; return false; }
        ++i_;
        skip_ws();
        if (i_ >= s_.size() || (s_[i_] != '"' && s_[i_] != '\'')) { err = "unquoted attribute in <" + n + ">"; return false; }
        char qc = s_[i_++];
        size_t e2 = s_.find(qc, i_);
        if (e2 == std::string::npos) { err = "unterminated attribute value in <" + n + ">"; return false; }
        e->attr[k] = s_.substr(i_, e2 - i_);
        i_ = e2 + 1;
      }
    }
    if (!stack.empty()) { err = "unclosed element <" + stack.back()->name + ">"; return false; }
    if (!have_root) { err = "no root element"; return false; }
    return true;
  }

 private:
  bool starts(const char* p) const { return s_.compare(i_, std::char_traits<char>::length(p), p) == 0; }
  void skip_ws() { while (i_ < s_.size() && std::isspace((unsigned char)s_[i_])) ++i_; }
  std::string ident() {
    size_t b = i_;
    while (i_ < s_.size() && (std::isalnum((unsigned char)s_[i_]) || s_[i_] == '_' || s_[i_] == ':' || s_[i_] == '-' || s_[i_] == '.')) ++i_;
    return s_.substr(b, i_ - b);
  }
  const std::string& s_;
  size_t i_ = 0;
};

bool parse_vec(const std::string& s, int n, double* out) {
  std::istringstream is(s);
  for (int i = 0; i < n; ++i)
    if (!(is >> out[i])) return false;
  double extra;
  return !(is >> extra);
}

struct Mat3 { double a[9]; };
Mat3 ident3() { return Mat3{{1, 0, 0, 0, 1, 0, 0, 0, 1}}; }
Mat3 mul(const Mat3& x, const Mat3& y) {
  Mat3 r;
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      double s = 0;
      for (int k = 0; k < 3; ++k) s += x.a[3 * i + k] * y.a[3 * k + j];
      r.a[3 * i + j] = s;
    }
  return r;
}
Mat3 transp(const Mat3& x) {
  Mat3 r;
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) r.a[3 * i + j] = x.a[3 * j + i];
  return r;
}
void mulv(const Mat3& x, const double* v, double* o) {
  for (int i = 0; i < 3; ++i) o[i] = x.a[3 * i] * v[0] + x.a[3 * i + 1] * v[1] + x.a[3 * i + 2] * v[2];
}
// URDF fixed-axis rpy: R = Rz(yaw) Ry(pitch) Rx(roll)
Mat3 rpy(double r, double p, double y) {
  double cr = std::cos(r), sr = std::sin(r), cp = std::cos(p), sp = std::sin(p), cy = std::cos(y), sy = std::sin(y);
  return Mat3{{cy * cp, cy * sp * sr - sy * cr, cy * sp * cr + sy * sr,
               sy * cp, sy * sp * sr + cy * cr, sy * sp * cr - cy * sr,
               -sp, cp * sr, cp * cr}};
}

struct Inertia {  // mass, COM and inertia about the COM, all in some body frame
  double m = 0, c[3] = {0, 0, 0};
  Mat3 I{{0, 0, 0, 0, 0, 0, 0, 0, 0}};
};

// lump b (given in a's frame) into a
void lump(Inertia& a, const Inertia& b) {
  if (b.m == 0.0) return;
  double m = a.m + b.m, c[3];
  for (int k = 0; k < 3; ++k) c[k] = (a.m * a.c[k] + b.m * b.c[k]) / m;
  Mat3 I{{0, 0, 0, 0, 0, 0, 0, 0, 0}};
  const Inertia* parts[2] = {&a, &b};
  for (const Inertia* p : parts) {
    double d[3] = {p->c[0] - c[0], p->c[1] - c[1], p->c[2] - c[2]};
    double dd = d[0] * d[0] + d[1] * d[1] + d[2] * d[2];
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) I.a[3 * i + j] += p->I.a[3 * i + j] + p->m * ((i == j ? dd : 0.0) - d[i] * d[j]);
  }
  a.m = m;
  for (int k = 0; k < 3; ++k) a.c[k] = c[k];
  a.I = I;
}

struct Joint {
  std::string name, type, parent, child;
  Mat3 R;
  double r[3], axis[3];
};

struct Body {
  int parent;
  Mat3 Rt;
  double rt[3], axis[3];
  Inertia in;
  std::string name;
};

struct Frame { int body; Mat3 R; double r[3]; };

}  // namespace

int load_urdf(const std::string& path, const std::vector<std::string>& foot_links, FlatModel& out, std::string& err) {
  std::ifstream f(path, std::ios::binary);
  if (!f) { err = "cannot open URDF '" + path + "'"; return WBC_E_IO; }
  std::stringstream ss;
  ss << f.rdbuf();
  const std::string text = ss.str();
  Elem root;
  {
    Xml x(text);
    if (!x.parse(root, err)) return WBC_E_PARSE;
  }
  if (root.name != "robot") { err = "root element is <" + root.name + ">, expected <robot>"; return WBC_E_PARSE; }

  std::map<std::string, Inertia> links;
  std::vector<std::string> order;
  std::vector<Joint> joints;
  for (auto& k : root.kids) {
    if (k->name == "link") {
      std::string n = k->get("name", "");
      if (n.empty() || links.count(n)) { err = "link without a unique name"; return WBC_E_PARSE; }
      Inertia in;
      if (const Elem* ine = k->child("inertial")) {
        double xyz[3] = {0, 0, 0}, r3[3] = {0, 0, 0};
        if (const Elem* o = ine->child("origin")) {
          if (!parse_vec(o->get("xyz", "0 0 0"), 3, xyz) || !parse_vec(o->get("rpy", "0 0 0"), 3, r3)) {
            err = "bad inertial origin in link " + n; return WBC_E_PARSE;
          }
        }
        const Elem* ma = ine->child("mass");
        const Elem* it = ine->child("inertia");
        if (!ma || !it) { err = "inertial without mass/inertia in link " + n; return WBC_E_PARSE; }
        double mv;
        if (!parse_vec(ma->get("value", ""), 1, &mv)) { err = "bad mass in link " + n; return WBC_E_PARSE; }
        const char* keys[6] = {"ixx", "ixy", "ixz", "iyy", "iyz", "izz"};
        double iv[6];
        for (int q = 0; q < 6; ++q)
          if (!parse_vec(it->get(keys[q], "0"), 1, &iv[q])) { err = "bad inertia in link " + n; return WBC_E_PARSE; }
        Mat3 I{{iv[0], iv[1], iv[2], iv[1], iv[3], iv[4], iv[2], iv[4], iv[5]}};
        Mat3 Ri = rpy(r3[0], r3[1], r3[2]);
        in.m = mv;
        for (int q = 0; q < 3; ++q) in.c[q] = xyz[q];
        in.I = mul(mul(Ri, I), transp(Ri));
      }
      links[n] = in;
      order.push_back(n);
    } else if (k->name == "joint") {
      Joint j;
      j.name = k->get("name", "");
      j.type = k->get("type", "");
      if (j.type != "revolute" && j.type != "continuous" && j.type != "fixed") {
        err = "unsupported joint type '" + j.type + "' (joint " + j.name + ")"; return WBC_E_PARSE;
      }
      const Elem* p = k->child("parent");
      const Elem* c = k->child("child");
      if (!p || !c) { err = "joint " + j.name + " lacks parent/child"; return WBC_E_PARSE; }
      j.parent = p->get("link", "");
      j.child = c->get("link", "");
      double xyz[3] = {0, 0, 0}, r3[3] = {0, 0, 0};
      if (const Elem* o = k->child("origin")) {
        if (!parse_vec(o->get("xyz", "0 0 0"), 3, xyz) || !parse_vec(o->get("rpy", "0 0 0"), 3, r3)) {
          err = "bad origin in joint " + j.name; return WBC_E_PARSE;
        }
      }
      j.R = rpy(r3[0], r3[1], r3[2]);
      for (int q = 0; q < 3; ++q) j.r[q] = xyz[q];
      double ax[3] = {1, 0, 0};
      if (const Elem* a = k->child("axis"))
        if (!parse_vec(a->get("xyz", "1 0 0"), 3, ax)) { err = "bad axis in joint " + j.name; return WBC_E_PARSE; }
      double nrm = std::sqrt(ax[0] * ax[0] + ax[1] * ax[1] + ax[2] * ax[2]);
      if (j.type != "fixed" && nrm == 0.0) { err = "zero axis in joint " + j.name; return WBC_E_PARSE; }
      for (int q = 0; q < 3; ++q) j.axis[q] = nrm > 0 ? ax[q] / nrm : ax[q];
      joints.push_back(j);
    }
  }
  for (auto& j : joints)
    if (!links.count(j.parent) || !links.count(j.child)) { err = "joint " + j.name + " references an unknown link"; return WBC_E_PARSE; }
  std::string rootlink;
  int nroots = 0;
  for (auto& n : order) {
    bool is_child = false;
    for (auto& j : joints) if (j.child == n) is_child = true;
    if (!is_child) { rootlink = n; ++nroots; }
  }
  if (nroots != 1) { err = "URDF must have exactly one root link"; return WBC_E_PARSE; }

  std::vector<Body> bodies;
  std::map<std::string, Frame> frames;
  std::vector<std::string> jnames;
  Body base;
  base.parent = -1; base.Rt = ident3();
  for (int q = 0; q < 3; ++q) base.rt[q] = base.axis[q] = 0;
  base.name = rootlink;
  bodies.push_back(base);

  // depth-first, children in document order (explicit stack would reorder; use recursion)
  struct Rec {
    std::vector<Body>& bodies; std::map<std::string, Frame>& frames; std::vector<std::string>& jnames;
    const std::map<std::string, Inertia>& links; const std::vector<Joint>& joints; int depth = 0; bool bad = false;
    void visit(const std::string& link, int body, const Mat3& R, const double* r) {
      if (++depth > 256) { bad = true; return; }
      Frame fr; fr.body = body; fr.R = R; for (int q = 0; q < 3; ++q) fr.r[q] = r[q];
      frames[link] = fr;
      const Inertia& li = links.at(link);
      Inertia t; t.m = li.m;
      double rc[3]; mulv(R, li.c, rc);
      for (int q = 0; q < 3; ++q) t.c[q] = r[q] + rc[q];
      t.I = mul(mul(R, li.I), transp(R));
      lump(bodies[body].in, t);
      for (auto& j : joints) {
        if (j.parent != link) continue;
        Mat3 Rj = mul(R, j.R);
        double rj[3], tmp[3];
        mulv(R, j.r, tmp);
        for (int q = 0; q < 3; ++q) rj[q] = r[q] + tmp[q];
        if (j.type == "fixed") {
          visit(j.child, body, Rj, rj);
        } else {
          Body b; b.parent = body; b.Rt = Rj;
          for (int q = 0; q < 3; ++q) { b.rt[q] = rj[q]; b.axis[q] = j.axis[q]; }
          b.name = j.child;
          bodies.push_back(b);
          jnames.push_back(j.name);
          double z[3] = {0, 0, 0};
          visit(j.child, (int)bodies.size() - 1, ident3(), z);
        }
      }
      --depth;
    }
  } rec{bodies, frames, jnames, links, joints};
  double z3[3] = {0, 0, 0};
  rec.visit(rootlink, 0, ident3(), z3);
  if (rec.bad) { err = "kinematic loop or tree deeper than 256"; return WBC_E_PARSE; }

  const int nb = (int)bodies.size();
  std::vector<std::string> feet = foot_links;
  if (feet.empty()) {
    for (int i = 1; i < nb; ++i) {
      bool leaf = true;
      for (auto& b : bodies) if (b.parent == i) leaf = false;
      if (!leaf) continue;
      std::string last;
      for (auto& n : order) if (frames.count(n) && frames[n].body == i) last = n;
      feet.push_back(last);
    }
  }
  out = FlatModel();
  out.nb = nb;
  for (auto& b : bodies) {
    out.parent.push_back(b.parent);
    for (int q = 0; q < 9; ++q) out.Rt.push_back(b.Rt.a[q]);
    for (int q = 0; q < 3; ++q) { out.rt.push_back(b.rt[q]); out.axis.push_back(b.axis[q]); out.com.push_back(b.in.c[q]); }
    out.mass.push_back(b.in.m);
    const double* I = b.in.I.a;
    const double six[6] = {I[0], I[1], I[2], I[4], I[5], I[8]};
    for (double x : six) out.Ic.push_back(x);
    out.body_names.push_back(b.name);
  }
  for (auto& n : feet) {
    if (!frames.count(n)) { err = "foot link '" + n + "' not found"; return WBC_E_PARSE; }
    out.foot_body.push_back(frames[n].body);
    for (int q = 0; q < 3; ++q) out.foot_off.push_back(frames[n].r[q]);
  }
  out.joint_names = jnames;
  out.foot_links = feet;
  return WBC_OK;
}

int quadruped_topology(const FlatModel& m, int leg_body[4][3], std::string& err) {
  if (m.nb != 13 || m.nf() != 4) {
    err = "HIP kernels need a floating base with 4 legs x 3 revolute joints and 4 feet (got " +
          std::to_string(m.nb - 1) + " joints, " + std::to_string(m.nf()) + " feet)";
    return WBC_E_TOPOLOGY;
  }
  bool used[13] = {false};
  for (int l = 0; l < 4; ++l) {
    int b = m.foot_body[l];
    for (int k = 2; k >= 0; --k) {
      if (b <= 0 || b >= m.nb || used[b]) { err = "foot " + std::to_string(l) + " is not at the end of its own 3-joint chain"; return WBC_E_TOPOLOGY; }
      used[b] = true;
      leg_body[l][k] = b;
      b = m.parent[b];
    }
    if (b != 0) { err = "leg " + std::to_string(l) + " has more than 3 joints"; return WBC_E_TOPOLOGY; }
  }
  return WBC_OK;
}

}  // namespace wbc
