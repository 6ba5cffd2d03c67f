// urdf_chain.hpp -- host-side model loading: URDF text -> flat kinematic chain.
//
// Restates KinematicChain::from_urdf / parse_urdf / urdf_to_tfm of
// /root/reference/crates/optik/src/kinematics.rs:18-105, 263-319 (which lean on
// the urdf-rs and petgraph crates) with a small self-contained XML reader and a
// breadth-first path search.  One-time setup; the GPU consumes its output table.
#pragma once

#include <cmath>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <map>
#include <stdexcept>
#include <string>
#include <vector>

namespace optik_host {

struct HPose {
    double t[3] = {0, 0, 0};
    double q[4] = {0, 0, 0, 1};  // [i, j, k, w]
};

inline void cross3(const double a[3], const double b[3], double o[3]) {
    const double x = a[1] * b[2] - a[2] * b[1];
    const double y = a[2] * b[0] - a[0] * b[2];
    const double z = a[0] * b[1] - a[1] * b[0];
    o[0] = x; o[1] = y; o[2] = z;
}

inline void qmul(const double a[4], const double b[4], double o[4]) {
    const double w = a[3] * b[3] - a[0] * b[0] - a[1] * b[1] - a[2] * b[2];
    const double i = a[3] * b[0] + a[0] * b[3] + a[1] * b[2] - a[2] * b[1];
    const double j = a[3] * b[1] - a[0] * b[2] + a[1] * b[3] + a[2] * b[0];
    const double k = a[3] * b[2] + a[0] * b[1] - a[1] * b[0] + a[2] * b[3];
    o[0] = i; o[1] = j; o[2] = k; o[3] = w;
}

inline void qrot(const double q[4], const double r[3], double o[3]) {
    double t[3], c[3];
    cross3(q, r, t);
    t[0] *= 2.0; t[1] *= 2.0; t[2] *= 2.0;
    cross3(q, t, c);
    const double x = t[0] * q[3] + c[0] + r[0];
    const double y = t[1] * q[3] + c[1] + r[1];
    const double z = t[2] * q[3] + c[2] + r[2];
    o[0] = x; o[1] = y; o[2] = z;
}

// Isometry3 product (t1 + q1 t2, q1 q2).
inline HPose pose_mul(const HPose &a, const HPose &b) {
    HPose o;
    double s[3];
    qrot(a.q, b.t, s);
    for (int i = 0; i < 3; ++i) o.t[i] = a.t[i] + s[i];
    qmul(a.q, b.q, o.q);
    return o;
}

inline bool pose_is_identity(const HPose &p) {
    return p.t[0] == 0 && p.t[1] == 0 && p.t[2] == 0 && p.q[0] == 0 && p.q[1] == 0 && p.q[2] == 0
           && p.q[3] == 1;
}

// urdf_to_tfm, kinematics.rs:263-267 (nalgebra from_euler_angles(roll, pitch, yaw)).
inline HPose pose_from_xyz_rpy(const double xyz[3], const double rpy[3]) {
    const double sr = std::sin(rpy[0] * 0.5), cr = std::cos(rpy[0] * 0.5);
    const double sp = std::sin(rpy[1] * 0.5), cp = std::cos(rpy[1] * 0.5);
    const double sy = std::sin(rpy[2] * 0.5), cy = std::cos(rpy[2] * 0.5);
    HPose o;
    o.q[3] = cr * cp * cy + sr * sp * sy;
    o.q[0] = sr * cp * cy - cr * sp * sy;
    o.q[1] = cr * sp * cy + sr * cp * sy;
    o.q[2] = cr * cp * sy - sr * sp * cy;
    o.t[0] = xyz[0]; o.t[1] = xyz[1]; o.t[2] = xyz[2];
    return o;
}

// ---- minimal XML reader --------------------------------------------------------

struct XmlNode {
    std::string name;
    std::map<std::string, std::string> attr;
    std::vector<XmlNode> children;
    const XmlNode *child(const char *n) const {
        for (const auto &c : children)
            if (c.name == n) return &c;
        return nullptr;
    }
};

class XmlReader {
   public:
    explicit XmlReader(const std::string &s) : s_(s) {}
    XmlNode parse_document() {
        skip_misc();
        XmlNode root = parse_element();
        return root;
    }

   private:
    const std::string &s_;
    size_t p_ = 0;
    [[noreturn]] void err(const char *what) const {
        throw std::runtime_error(std::string("error parsing URDF file! (") + what + " at byte "
                                 + std::to_string(p_) + ")");
    }
    bool starts(const char *lit) const { return s_.compare(p_, std::strlen(lit), lit) == 0; }
    void skip_ws() {
        while (p_ < s_.size() && std::isspace((unsigned char)s_[p_])) ++p_;
    }
    void skip_until(const char *lit) {
        const size_t e = s_.find(lit, p_);
        if (e == std::string::npos) err("unterminated construct");
        p_ = e + std::strlen(lit);
    }
    void skip_misc() {  // whitespace, <?...?>, <!-- -->, <!DOCTYPE ...>
        for (;;) {
            skip_ws();
            if (starts("<?")) skip_until("?>");
            else if (starts("<!--")) skip_until("-->");
            else if (starts("<!")) skip_until(">");
            else return;
        }
    }
    std::string parse_name() {
        const size_t b = p_;
        while (p_ < s_.size()
               && (std::isalnum((unsigned char)s_[p_]) || s_[p_] == '_' || s_[p_] == ':' || s_[p_] == '-'
                   || s_[p_] == '.'))
            ++p_;
        if (p_ == b) err("expected a name");
        return s_.substr(b, p_ - b);
    }
    XmlNode parse_element() {
        if (p_ >= s_.size() || s_[p_] != '<') err("expected '<'");
        ++p_;
        XmlNode n;
        n.name = parse_name();
        for (;;) {
            skip_ws();
            if (p_ >= s_.size()) err("unterminated tag");
            if (starts("/>")) { p_ += 2; return n; }
            if (s_[p_] == '>') { ++p_; break; }
            const std::string key = parse_name();
            skip_ws();
            if (p_ >= s_.size() || s_[p_] != '=') err("expected '='");
            ++p_;
            skip_ws();
            if (p_ >= s_.size() || (s_[p_] != '"' && s_[p_] != '\'')) err("expected a quoted value");
            const char quote = s_[p_++];
            const size_t e = s_.find(quote, p_);
            if (e == std::string::npos) err("unterminated attribute");
            n.attr[key] = s_.substr(p_, e - p_);
            p_ = e + 1;
        }
        for (;;) {  // content
            const size_t lt = s_.find('<', p_);
            if (lt == std::string::npos) err("unterminated element");
            p_ = lt;
            if (starts("<!--")) { skip_until("-->"); continue; }
            if (starts("<![CDATA[")) { skip_until("]]>"); continue; }
            if (starts("<?")) { skip_until("?>"); continue; }
            if (starts("</")) {
                p_ += 2;
                const std::string close = parse_name();
                if (close != n.name) err("mismatched closing tag");
                skip_ws();
                if (p_ >= s_.size() || s_[p_] != '>') err("expected '>'");
                ++p_;
                return n;
            }
            n.children.push_back(parse_element());
        }
    }
};

inline void parse_floats(const XmlNode *n, const char *key, int cnt, const double *dflt, double *out) {
    for (int i = 0; i < cnt; ++i) out[i] = dflt[i];
    if (!n) return;
    auto it = n->attr.find(key);
    if (it == n->attr.end()) return;
    const char *c = it->second.c_str();
    for (int i = 0; i < cnt; ++i) {
        char *end = nullptr;
        const double v = std::strtod(c, &end);
        if (end == c) throw std::runtime_error("error parsing URDF file! (bad number list)");
        out[i] = v;
        c = end;
    }
}

// ---- chain --------------------------------------------------------------------

enum JointKind { FIXED = 0, REVOLUTE = 1, PRISMATIC = 2 };

struct ChainJoint {
    std::string name;
    int kind = FIXED;
    double axis[3] = {0, 0, 0};
    bool has_limit = false;  // articulated joints carry one (lower, upper) pair
    double lower = 0, upper = 0;
    HPose origin;
};

struct Chain {
    std::vector<ChainJoint> joints;  // kinematics.rs:8-10
    int num_positions() const {
        int n = 0;
        for (const auto &j : joints) n += (j.kind != FIXED);
        return n;
    }
};

// KinematicChain::from_urdf, kinematics.rs:18-105.  Throws std::runtime_error with
// the reference's panic messages.
inline Chain chain_from_urdf(const std::string &urdf, const std::string &base_link,
                             const std::string &ee_link) {
    XmlReader rd(urdf);
    const XmlNode root = rd.parse_document();
    if (root.name != "robot") throw std::runtime_error("error parsing URDF file! (no <robot> element)");

    // parse_urdf, kinematics.rs:269-319
    std::vector<std::string> links;
    for (const auto &c : root.children)
        if (c.name == "link") {
            auto it = c.attr.find("name");
            if (it == c.attr.end()) throw std::runtime_error("error parsing URDF file! (link without a name)");
            links.push_back(it->second);
        }
    auto link_index = [&](const std::string &name) -> int {
        for (size_t i = 0; i < links.size(); ++i)
            if (links[i] == name) return (int)i;
        return -1;
    };
    struct Edge { int parent, child; ChainJoint joint; };
    std::vector<Edge> edges;
    const double zero3[3] = {0, 0, 0}, xaxis[3] = {1, 0, 0};
    for (const auto &c : root.children) {
        if (c.name != "joint") continue;
        Edge e;
        auto nm = c.attr.find("name");
        e.joint.name = nm != c.attr.end() ? nm->second : "";
        auto ty = c.attr.find("type");
        if (ty == c.attr.end()) throw std::runtime_error("error parsing URDF file! (joint without a type)");
        const XmlNode *par = c.child("parent"), *chi = c.child("child");
        if (!par || !chi || !par->attr.count("link") || !chi->attr.count("link"))
            throw std::runtime_error("error parsing URDF file! (joint without parent/child)");
        const std::string &pl = par->attr.at("link"), &cl = chi->attr.at("link");
        e.parent = link_index(pl);
        if (e.parent < 0) throw std::runtime_error("joint parent link '" + pl + "' does not exist");
        e.child = link_index(cl);
        if (e.child < 0) throw std::runtime_error("joint child link '" + cl + "' does not exist");
        double ax[3];
        parse_floats(c.child("axis"), "xyz", 3, xaxis, ax);  // urdf-rs default axis (1, 0, 0)
        if (ty->second == "revolute") e.joint.kind = REVOLUTE;
        else if (ty->second == "prismatic") e.joint.kind = PRISMATIC;
        else if (ty->second == "fixed") e.joint.kind = FIXED;
        else throw std::runtime_error("joint type not supported: " + ty->second);
        if (e.joint.kind != FIXED) {
            const double nrm = std::sqrt(ax[0] * ax[0] + ax[1] * ax[1] + ax[2] * ax[2]);  // Unit::new_normalize
            for (int i = 0; i < 3; ++i) e.joint.axis[i] = ax[i] / nrm;
        }
        double lower = 0, upper = 0;
        parse_floats(c.child("limit"), "lower", 1, &lower, &lower);
        parse_floats(c.child("limit"), "upper", 1, &upper, &upper);
        e.joint.has_limit = true;
        if (upper - lower > 0.0) { e.joint.lower = lower; e.joint.upper = upper; }  // kinematics.rs:299-303
        else { e.joint.lower = -INFINITY; e.joint.upper = INFINITY; }
        double xyz[3], rpy[3];
        parse_floats(c.child("origin"), "xyz", 3, zero3, xyz);
        parse_floats(c.child("origin"), "rpy", 3, zero3, rpy);
        e.joint.origin = pose_from_xyz_rpy(xyz, rpy);
        edges.push_back(e);
    }

    // assert!(!is_cyclic_directed(&graph)), kinematics.rs:21 (Kahn's algorithm)
    {
        std::vector<int> indeg(links.size(), 0);
        for (const auto &e : edges) indeg[e.child]++;
        std::deque<int> q;
        for (size_t i = 0; i < links.size(); ++i)
            if (indeg[i] == 0) q.push_back((int)i);
        size_t seen = 0;
        while (!q.empty()) {
            const int u = q.front();
            q.pop_front();
            ++seen;
            for (const auto &e : edges)
                if (e.parent == u && --indeg[e.child] == 0) q.push_back(e.child);
        }
        if (seen != links.size()) throw std::runtime_error("robot model contains loops");
    }
    const int base = link_index(base_link);
    if (base < 0) throw std::runtime_error("base link '" + base_link + "' does not exist");
    const int ee = link_index(ee_link);
    if (ee < 0) throw std::runtime_error("EE link '" + ee_link + "' does not exist");

    // A* with unit edge cost and zero heuristic (kinematics.rs:35-42) == breadth-first search
    std::vector<int> via(links.size(), -2);  // edge used to reach the link
    via[base] = -1;
    std::deque<int> bfs{base};
    while (!bfs.empty() && via[ee] == -2) {
        const int u = bfs.front();
        bfs.pop_front();
        for (size_t k = 0; k < edges.size(); ++k)
            if (edges[k].parent == u && via[edges[k].child] == -2) {
                via[edges[k].child] = (int)k;
                bfs.push_back(edges[k].child);
            }
    }
    if (via[ee] == -2) throw std::runtime_error("no path from base to EE link");
    std::vector<int> path;
    for (int u = ee; via[u] >= 0; u = edges[via[u]].parent) path.insert(path.begin(), via[u]);

    // fold fixed joints into the next articulated joint's origin (kinematics.rs:64-86;
    // the multiplication order is the reference's, quirk Q1)
    Chain chain;
    HPose collapsed;
    for (int k : path) {
        const ChainJoint &j = edges[k].joint;
        if (j.kind == FIXED) {
            collapsed = pose_mul(j.origin, collapsed);
        } else {
            ChainJoint nj = j;
            nj.origin = pose_mul(j.origin, collapsed);
            chain.joints.push_back(nj);
            collapsed = HPose();
        }
    }
    if (!pose_is_identity(collapsed)) {  // trailing fixed joints, kinematics.rs:90-97
        ChainJoint tip;
        tip.kind = FIXED;
        tip.has_limit = false;
        tip.origin = collapsed;
        chain.joints.push_back(tip);
    }
    if (chain.num_positions() <= 0) throw std::runtime_error("kinematic chain is empty");
    return chain;
}

}  // namespace optik_host
