#include "pdbqt.h"

#include <algorithm>
#include <cctype>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <iomanip>
#include <memory>
#include <fstream>
#include <sstream>
#include <stdexcept>

#include "../../include/mi_gnina.h"

namespace gnina_amd {
namespace {

// smina types (atom_constants.h:45-133): AutoDock name, covalent radius, "heteroatom" flag.  Index = smt.
struct TypeRow {
  const char *smina, *ad;
  float covalent;
  bool hetero;
};
const TypeRow kTypes[28] = {
    {"Hydrogen", "H", 0.37f, false},
    {"PolarHydrogen", "HD", 0.37f, false},
    {"AliphaticCarbonXSHydrophobe", "C", 0.77f, false},
    {"AliphaticCarbonXSNonHydrophobe", "C", 0.77f, false},
    {"AromaticCarbonXSHydrophobe", "A", 0.77f, false},
    {"AromaticCarbonXSNonHydrophobe", "A", 0.77f, false},
    {"Nitrogen", "N", 0.75f, true},
    {"NitrogenXSDonor", "N", 0.75f, true},
    {"NitrogenXSDonorAcceptor", "NA", 0.75f, true},
    {"NitrogenXSAcceptor", "NA", 0.75f, true},
    {"Oxygen", "O", 0.73f, true},
    {"OxygenXSDonor", "O", 0.73f, true},
    {"OxygenXSDonorAcceptor", "OA", 0.73f, true},
    {"OxygenXSAcceptor", "OA", 0.73f, true},
    {"Sulfur", "S", 1.02f, true},
    {"SulfurAcceptor", "SA", 1.02f, true},
    {"Phosphorus", "P", 1.06f, true},
    {"Fluorine", "F", 0.71f, true},
    {"Chlorine", "Cl", 0.99f, true},
    {"Bromine", "Br", 1.14f, true},
    {"Iodine", "I", 1.33f, true},
    {"Magnesium", "Mg", 1.30f, true},
    {"Manganese", "Mn", 1.39f, true},
    {"Zinc", "Zn", 1.31f, true},
    {"Calcium", "Ca", 1.74f, true},
    {"Iron", "Fe", 1.25f, true},
    {"GenericMetal", "M", 1.75f, true},
    {"Boron", "B", 0.90f, false},
};
enum { kH = 0, kHD = 1, kGenericMetal = 26, kNumTypes = 28 };

float max_covalent_radius() {
  float m = 0;
  for (const TypeRow &r : kTypes) m = std::max(m, r.covalent);
  return m;
}

// string_to_smina_type (atom_constants.h:227-253): <= 2 characters = AutoDock name (first row that carries it),
// "Se" counts as "S", anything else short is a generic metal; longer strings are full smina names
int string_to_smina_type(const std::string &name) {
  if (name.empty()) return kNumTypes;
  if (name.size() <= 2) {
    for (int i = 0; i < kNumTypes; i++)
      if (name == kTypes[i].ad) return i;
    if (name == "Se") return string_to_smina_type("S");
    return kGenericMetal;
  }
  for (int i = 0; i < kNumTypes; i++)
    if (name == kTypes[i].smina) return i;
  return kNumTypes;
}

// adjust_smina_type (atom_constants.h:280-309)
int adjust_smina_type(int t, bool h_bonded, bool hetero_bonded) {
  switch (t) {
    case 2: case 3: return hetero_bonded ? 3 : 2;
    case 4: case 5: return hetero_bonded ? 5 : 4;
    case 6: case 7: return h_bonded ? 7 : 6;
    case 8: case 9: return h_bonded ? 8 : 9;
    case 10: case 11: return h_bonded ? 11 : 10;
    case 12: case 13: return h_bonded ? 12 : 13;
    default: return t;
  }
}

struct PAtom {
  unsigned number = 0;
  float c[3] = {0, 0, 0};
  int sm = 0;
  unsigned line = 0;  // 1-based line of the ATOM record in the input
};

[[noreturn]] void fail(const std::string &name, unsigned line, const std::string &what) {
  throw std::runtime_error(name + ":" + std::to_string(line) + ": " + what);
}

bool starts_with(const std::string &s, const char *p) { return s.compare(0, std::strlen(p), p) == 0; }

// parse_pdbqt_atom_string (parse_pdbqt.cpp:103-122): 1-based columns 7-11 number, 31-38 / 39-46 / 47-54
// coordinates, 78.. the AutoDock type (omit_whitespace(str, 78, 79) runs to the end of the line)
template <typename T> T field(const std::string &name, unsigned line, const std::string &s, size_t i, size_t j, const char *what) {
  if (j > s.size()) fail(name, line, "ATOM syntax incorrect: The line is too short");
  std::istringstream is(s.substr(i - 1, j - i + 1));
  T v;
  is >> v;
  std::string rest;
  if (!is || (is >> rest)) fail(name, line, std::string("ATOM syntax incorrect: \"") + s.substr(i - 1, j - i + 1) + "\" is not a valid " + what);
  return v;
}

PAtom parse_atom(const std::string &name, unsigned line, const std::string &s) {
  PAtom a;
  a.line = line;
  const long num = field<long>(name, line, s, 7, 11, "atom number");
  if (num < 0) fail(name, line, "ATOM syntax incorrect: negative atom number");
  a.number = (unsigned)num;
  a.c[0] = field<float>(name, line, s, 31, 38, "coordinate");
  a.c[1] = field<float>(name, line, s, 39, 46, "coordinate");
  a.c[2] = field<float>(name, line, s, 47, 54, "coordinate");
  if (s.size() < 78) fail(name, line, "ATOM syntax incorrect: The line is too short");
  size_t b = 77, e = s.size();
  while (b < e && std::isspace((unsigned char)s[b])) b++;
  while (e > b && std::isspace((unsigned char)s[e - 1])) e--;
  const std::string ad = s.substr(b, e - b);
  a.sm = string_to_smina_type(ad);
  if (a.sm >= kNumTypes)
    fail(name, line, "ATOM syntax incorrect: \"" + ad + "\" is not a valid AutoDock type. Note that AutoDock atom types are case-sensitive.");
  return a;
}

bool ignorable(const std::string &s) {
  return s.empty() || starts_with(s, "WARNING") || starts_with(s, "REMARK") || starts_with(s, "USER") || starts_with(s, "TER");
}

// ---- covalent bonds + type adjustment (model::assign_bonds / assign_types, model.cpp:560-655) -----------------
// mob(i, j): 0 variable, 1 fixed, 2 rotor.  A cell list finds the candidates (the same set the reference's 15 A
// "beads" give: every atom within the cut-off is seen); they are then visited in the reference's order -- bead by
// bead, index order inside a bead (model.cpp:540-557,585-602) -- because the order of an atom's bond list feeds the
// order-dependent depth-first walk of bonded_to() that decides which 1-4 pairs are excluded.
template <typename Mob>
void bonds_and_types(const std::vector<PAtom> &atoms, Mob mob, std::vector<std::vector<int>> &bonds, std::vector<int> &sm_out) {
  const int n = (int)atoms.size();
  const float allow = 1.1f, maxcov = max_covalent_radius();
  const float cell = allow * 2 * maxcov;
  bonds.assign(n, {});
  // cell list
  float lo[3] = {1e30f, 1e30f, 1e30f};
  for (const PAtom &a : atoms)
    for (int k = 0; k < 3; k++) lo[k] = std::min(lo[k], a.c[k]);
  auto key = [&](const float *c, int *q) {
    for (int k = 0; k < 3; k++) q[k] = (int)std::floor((c[k] - lo[k]) / cell);
  };
  struct Entry {
    long long h;
    int i;
  };
  auto hash = [](const int *q) { return ((long long)q[0] * 2097152 + q[1]) * 2097152 + q[2]; };
  std::vector<Entry> ent(n);
  for (int i = 0; i < n; i++) {
    int q[3];
    key(atoms[i].c, q);
    ent[i] = {hash(q), i};
  }
  std::sort(ent.begin(), ent.end(), [](const Entry &a, const Entry &b) { return a.h < b.h || (a.h == b.h && a.i < b.i); });
  auto d2 = [&](int a, int b) {
    float s = 0;
    for (int k = 0; k < 3; k++) s += (atoms[a].c[k] - atoms[b].c[k]) * (atoms[a].c[k] - atoms[b].c[k]);
    return s;
  };
  // beads::add: an atom joins the first bead whose centre (that bead's first atom) is closer than 15 A
  std::vector<int> bead_of(n), bead_centre;
  for (int i = 0; i < n; i++) {
    int b = -1;
    for (size_t k = 0; k < bead_centre.size() && b < 0; k++)
      if (d2(i, bead_centre[k]) < 15.f * 15.f) b = (int)k;
    if (b < 0) {
      b = (int)bead_centre.size();
      bead_centre.push_back(i);
    }
    bead_of[i] = b;
  }
  std::vector<int> rel;
  for (int i = 0; i < n; i++) {
    const float ci = kTypes[atoms[i].sm].covalent;
    const float cut = allow * (ci + maxcov);
    // relevant atoms: not variable relative to i and within the generous cut-off, ascending index
    rel.clear();
    int q[3];
    key(atoms[i].c, q);
    for (int dx = -1; dx <= 1; dx++)
      for (int dy = -1; dy <= 1; dy++)
        for (int dz = -1; dz <= 1; dz++) {
          const int qq[3] = {q[0] + dx, q[1] + dy, q[2] + dz};
          const long long h = hash(qq);
          auto it = std::lower_bound(ent.begin(), ent.end(), h, [](const Entry &e, long long v) { return e.h < v; });
          for (; it != ent.end() && it->h == h; ++it) {
            const int j = it->i;
            if (j != i && mob(i, j) != 0 && d2(i, j) < cut * cut) rel.push_back(j);
          }
        }
    std::sort(rel.begin(), rel.end(), [&](int a, int b) {
      return bead_of[a] < bead_of[b] || (bead_of[a] == bead_of[b] && a < b);
    });
    for (int j : rel) {
      if (j <= i) continue;
      const float len = ci + kTypes[atoms[j].sm].covalent;  // optimal_covalent_bond_length
      const float r2 = d2(i, j);
      if (!(r2 < (allow * len) * (allow * len))) continue;
      // atom_exists_between: a heavy atom, immobile relative to both, closer to both than they are to each other
      bool blocked = false;
      for (int c : rel) {
        if (c == i || c == j || atoms[c].sm <= kHD) continue;
        if (mob(i, c) != 0 && mob(j, c) != 0 && d2(i, c) < r2 && d2(j, c) < r2) {
          blocked = true;
          break;
        }
      }
      if (!blocked) {
        bonds[i].push_back(j);
        bonds[j].push_back(i);
      }
    }
  }
  sm_out.resize(n);
  for (int i = 0; i < n; i++) {
    bool hd = false, het = false;
    for (int j : bonds[i]) {
      if (atoms[j].sm == kHD) hd = true;
      if (kTypes[atoms[j].sm].hetero) het = true;
    }
    sm_out[i] = adjust_smina_type(atoms[i].sm, hd, het);
  }
}

// ---- ligand structure ---------------------------------------------------------------------------------------
struct PS;
struct Node {
  PAtom a;
  std::vector<PS> ps;
};
struct PS {
  int immobile = -1;                  // which of `atoms` is the branch's first ("immobile") atom
  int axis_begin = -1, axis_end = -1; // model indices of the rotatable bond's two atoms
  std::vector<Node> atoms;
  bool essentially_empty() const {    // parsing.h:204-211
    for (size_t i = 0; i < atoms.size(); i++) {
      if (immobile >= 0 && (size_t)immobile != i) return false;
      if (!atoms[i].ps.empty()) return false;
    }
    return true;
  }
};

struct LineReader {
  std::istringstream in;
  unsigned count = 0;
  const std::string &name;
  LineReader(const std::string &n, const std::string &text) : in(text), name(n) {}
  std::vector<std::string> lines;
  bool next(std::string &s) {
    if (!std::getline(in, s)) return false;
    if (!s.empty() && s.back() == '\r') s.pop_back();
    count++;
    lines.push_back(s);
    return true;
  }
};

void two_unsigneds(LineReader &r, const std::string &s, const char *tag, unsigned &a, unsigned &b) {
  std::istringstream is(s.substr(std::strlen(tag)));
  long x, y;
  is >> x >> y;
  if (!is || x < 0 || y < 0) fail(r.name, r.count, "Syntax error");
  a = (unsigned)x, b = (unsigned)y;
}

void parse_branch(LineReader &r, PS &p, unsigned from, unsigned to);

void branch_aux(LineReader &r, const std::string &s, PS &p) {  // parse_pdbqt.cpp:249-271
  unsigned first, second;
  two_unsigneds(r, s, "BRANCH", first, second);
  for (Node &nd : p.atoms)
    if (nd.a.number == first) {
      PS branch;
      parse_branch(r, branch, first, second);
      nd.ps.push_back(std::move(branch));  // (fix_hydrogens is off: hydrogen-only branches stay branches)
      return;
    }
  fail(r.name, r.count, "No atom number " + std::to_string(first) + " in this branch");
}

void parse_branch(LineReader &r, PS &p, unsigned from, unsigned to) {  // parse_pdbqt.cpp:481-523
  std::string s;
  while (r.next(s)) {
    if (ignorable(s)) continue;
    if (starts_with(s, "BRANCH")) {
      branch_aux(r, s, p);
    } else if (starts_with(s, "ENDBRANCH")) {
      unsigned a, b;
      two_unsigneds(r, s, "ENDBRANCH", a, b);
      if (a != from || b != to) fail(r.name, r.count, "Inconsistent branch numbers");
      if (p.immobile < 0) fail(r.name, r.count, "Atom " + std::to_string(to) + " has not been found in this branch");
      return;
    } else if (starts_with(s, "ATOM  ") || starts_with(s, "HETATM")) {
      Node nd;
      nd.a = parse_atom(r.name, r.count, s);
      if (nd.a.number == to) p.immobile = (int)p.atoms.size();
      p.atoms.push_back(std::move(nd));
    } else if (starts_with(s, "MODEL")) {
      fail(r.name, r.count, "Unexpected multi-MODEL input. Use \"vina_split\" first?");
    } else {
      fail(r.name, r.count, "Unknown or inappropriate tag");
    }
  }
  // Input ended inside the BRANCH.  The reference's loop just ends (parse_pdbqt.cpp:481-523) and the branch is kept
  // as read; only a branch that has atoms but never met its second atom trips VINA_CHECK(immobile_atom) later
  // (parsing.h:182-202; an empty branch is skipped there).
  if (p.immobile < 0 && !p.atoms.empty())
    fail(r.name, r.count, "Unexpected end of file in BRANCH " + std::to_string(from) + " " + std::to_string(to) +
                              ": atom " + std::to_string(to) + " has not been found");
}

struct Builder {
  std::vector<PAtom> atoms;           // model order
  std::vector<float> local;           // [n][3]
  std::vector<int> node_of;
  std::vector<int> parent, abeg, aend;
  std::vector<float> rel_origin, rel_axis, origin;  // per node
  std::vector<std::vector<unsigned char>> fixed;    // sparse notes applied after the atom count is known
  struct Mark {
    int a, b;
    unsigned char t;
  };
  std::vector<Mark> marks;

  void insert(Node &nd, const float *frame_origin, int node) {  // parsing.h:151-158
    for (PS &c : nd.ps) c.axis_begin = (int)atoms.size();
    atoms.push_back(nd.a);
    for (int k = 0; k < 3; k++) local.push_back(nd.a.c[k] - frame_origin[k]);
    node_of.push_back(node);
  }
  void insert_immobiles(Node &nd, const float *frame_origin, int node) {  // parsing.h:159-163,194-202
    for (PS &c : nd.ps)
      if (!c.atoms.empty()) {
        if (c.immobile < 0) throw std::runtime_error("branch without its immobile atom");  // VINA_CHECK(immobile_atom)
        c.axis_end = (int)atoms.size();
        insert(c.atoms[c.immobile], frame_origin, node);
      }
  }
  // postprocess_branch (parse_pdbqt.cpp:346-382)
  void branch(PS &p, int node) {
    const float *org = &origin[3 * node];
    abeg[node] = (int)atoms.size();
    for (size_t i = 0; i < p.atoms.size(); i++) {
      Node &nd = p.atoms[i];
      if (!(p.immobile >= 0 && (size_t)p.immobile == i)) insert(nd, org, node);
      insert_immobiles(nd, org, node);
    }
    aend[node] = (int)atoms.size();
    for (int i = abeg[node]; i < aend[node]; i++) {
      if (p.axis_begin >= 0) marks.push_back({p.axis_begin, i, 1});
      if (p.axis_end >= 0) marks.push_back({p.axis_end, i, 1});
      for (int j = i + 1; j < aend[node]; j++) marks.push_back({i, j, 1});
    }
    if (p.axis_begin >= 0 && p.axis_end >= 0) marks.push_back({p.axis_begin, p.axis_end, 2});
    for (Node &nd : p.atoms)
      for (PS &c : nd.ps)
        if (!c.essentially_empty()) {
          // segment(origin = the branch's immobile atom, axis root = the parent-side atom): tree.h:152-203
          const float *o = c.atoms[c.immobile].a.c, *root = nd.a.c;
          float ax[3] = {o[0] - root[0], o[1] - root[1], o[2] - root[2]};
          const float nrm = std::sqrt(ax[0] * ax[0] + ax[1] * ax[1] + ax[2] * ax[2]);
          if (!(nrm >= 1.1920928955078125e-07f)) throw std::runtime_error("rotatable bond of zero length");
          const int k = (int)parent.size();
          parent.push_back(node);
          abeg.push_back(0);
          aend.push_back(0);
          for (int d = 0; d < 3; d++) {
            origin.push_back(o[d]);
            rel_origin.push_back(o[d] - origin[3 * node + d]);
            rel_axis.push_back((1 / nrm) * ax[d]);
          }
          branch(c, k);
        }
  }
};

}  // namespace

PdbqtReceptor parse_pdbqt_receptor(const std::string &name, const std::string &text) {  // parse_pdbqt_rigid, :145-183
  LineReader r(name, text);
  std::vector<PAtom> atoms;
  std::string s;
  while (r.next(s)) {
    if (starts_with(s, "ATOM  ") || starts_with(s, "HETATM")) atoms.push_back(parse_atom(name, r.count, s));
    else if (starts_with(s, "MODEL")) fail(name, r.count, "Unexpected multi-MODEL input. Use \"vina_split\" first?");
    // everything else is ignored ("let's be forgiving")
  }
  std::vector<std::vector<int>> bonds;
  std::vector<int> sm;
  bonds_and_types(atoms, [](int, int) { return 1; }, bonds, sm);  // rigid: every distance is fixed
  PdbqtReceptor out;
  for (size_t i = 0; i < atoms.size(); i++) {
    out.xyz.insert(out.xyz.end(), atoms[i].c, atoms[i].c + 3);
    out.smt.push_back(sm[i]);
  }
  return out;
}

// ROOT ... ENDROOT, then BRANCH blocks, up to TORSDOF / end of input (a ligand) or END_RES (one flexible residue):
// parse_pdbqt_root + parse_pdbqt_aux (parse_pdbqt.cpp:219-244,273-305).  Returns TORSDOF (-1: none seen).
static int parse_tree(LineReader &r, PS &root, bool residue) {
  const std::string &name = r.name;
  std::string s;
  bool have_root = false;
  while (!have_root && r.next(s)) {
    if (ignorable(s)) continue;
    if (starts_with(s, "ROOT")) {
      while (r.next(s)) {
        if (ignorable(s)) continue;
        if (starts_with(s, "ATOM  ") || starts_with(s, "HETATM")) {
          Node nd;
          nd.a = parse_atom(name, r.count, s);
          root.atoms.push_back(std::move(nd));
        } else if (starts_with(s, "ENDROOT")) {
          have_root = true;
          break;
        } else if (starts_with(s, "MODEL")) {
          fail(name, r.count, "Unexpected multi-MODEL input. Use \"vina_split\" first?");
        } else {
          fail(name, r.count, "Unknown or inappropriate tag");
        }
      }
    } else if (starts_with(s, "MODEL")) {
      fail(name, r.count, "Unexpected multi-MODEL input. Use \"vina_split\" first?");
    } else {
      fail(name, r.count, "Unknown or inappropriate tag");
    }
  }
  int torsdof = -1;
  while (r.next(s)) {
    if (ignorable(s)) continue;
    if (starts_with(s, "BRANCH")) {
      branch_aux(r, s, root);
    } else if (!residue && starts_with(s, "TORSDOF")) {
      if (torsdof >= 0) fail(name, r.count, "TORSDOF can occur only once");
      std::istringstream is(s.substr(7));
      long t;
      is >> t;
      if (!is || t < 0) fail(name, r.count, "Syntax error");
      torsdof = (int)t;
    } else if (residue && starts_with(s, "END_RES")) {
      return torsdof;
    } else if (starts_with(s, "MODEL")) {
      fail(name, r.count, "Unexpected multi-MODEL input. Use \"vina_split\" first?");
    } else {
      fail(name, r.count, "Unknown or inappropriate tag");
    }
  }
  return torsdof;  // a residue without END_RES ends with the input, as in the reference (parse_pdbqt.cpp:273-305)
}

// postprocess_ligand / postprocess_residue: the tree in model order.  Node 0 holds the ROOT atoms and the first
// ("immobile") atoms of the top-level branches -- the rigid root of a ligand, the inflex atoms of a residue.
static void build_tree(PS &root, Builder &b) {
  b.parent.push_back(-1);
  b.abeg.push_back(0);
  b.aend.push_back(0);
  for (int d = 0; d < 3; d++) {
    b.origin.push_back(root.atoms[0].a.c[d]);
    b.rel_origin.push_back(0.f);
    b.rel_axis.push_back(0.f);
  }
  b.branch(root, 0);
}

static std::vector<unsigned char> mobility_of(const Builder &b) {
  const int n = (int)b.atoms.size();
  std::vector<unsigned char> mobm((size_t)n * n, 0);
  for (const Builder::Mark &m : b.marks) {
    const int lo = std::min(m.a, m.b), hi = std::max(m.a, m.b);
    mobm[(size_t)lo * n + hi] = m.t;  // later marks overwrite earlier ones, like the reference's assignments
  }
  return mobm;
}

PdbqtLigand parse_pdbqt_ligand(const std::string &name, const std::string &text) {
  LineReader r(name, text);
  PS root;
  const int torsdof = parse_tree(r, root, false);
  if (root.atoms.empty()) fail(name, r.count, "No atoms in the ligand");
  if (torsdof < 0) fail(name, r.count, "Missing TORSDOF");

  // postprocess_ligand (:384-391): root frame at the first root atom
  Builder b;
  build_tree(root, b);
  const int n = (int)b.atoms.size();
  const std::vector<unsigned char> mobm = mobility_of(b);
  auto mob = [&](int i, int j) -> int { return i == j ? 1 : mobm[(size_t)std::min(i, j) * n + std::max(i, j)]; };
  std::vector<std::vector<int>> bonds;
  std::vector<int> sm;
  bonds_and_types(b.atoms, mob, bonds, sm);

  PdbqtLigand L;
  L.torsdof = torsdof;
  for (int i = 0; i < n; i++) {
    L.xyz.insert(L.xyz.end(), b.atoms[i].c, b.atoms[i].c + 3);
    L.smt.push_back(sm[i]);
    L.serial.push_back((int32_t)b.atoms[i].number);
  }
  L.local_xyz = b.local;
  L.node_parent.assign(b.parent.begin(), b.parent.end());
  L.node_atom_begin.assign(b.abeg.begin(), b.abeg.end());
  L.node_atom_end.assign(b.aend.begin(), b.aend.end());
  L.node_rel_origin = b.rel_origin;
  L.node_rel_axis = b.rel_axis;
  // initialize_pairs (model.cpp:682-703): variable distance, not within 3 bonds, both heavy
  for (int i = 0; i < n; i++) {
    // model::bonded_to(i, 3) (model.cpp:664-680): depth-first, an atom already listed is not expanded again
    struct Rec {
      static void go(int a, int depth, const std::vector<std::vector<int>> &bonds, std::vector<int> &out) {
        if (std::find(out.begin(), out.end(), a) != out.end()) return;
        out.push_back(a);
        if (depth > 0)
          for (int nb : bonds[a]) go(nb, depth - 1, bonds, out);
      }
    };
    std::vector<int> near;
    Rec::go(i, 3, bonds, near);
    for (int j = i + 1; j < n; j++) {
      if (mob(i, j) != 0) continue;
      if (std::find(near.begin(), near.end(), j) != near.end()) continue;
      if (sm[i] <= kHD || sm[j] <= kHD) continue;
      L.pairs.push_back(i);
      L.pairs.push_back(j);
    }
  }
  // conf_independent_inputs::num_tors (terms.cpp:39-106): half a count per heavy atom and rotatable bond to a heavy
  // neighbour that has another heavy neighbour (fix_hydrogens is off: the atom's own heavy degree is not checked)
  auto heavy_degree = [&](int a) {
    int k = 0;
    for (int nb : bonds[a]) k += sm[nb] > kHD;
    return k;
  };
  L.num_tors = 0.f;
  for (int i = 0; i < n; i++) {
    if (sm[i] <= kHD) continue;
    unsigned rotors = 0;
    for (int nb : bonds[i])
      if (mob(i, nb) == 2 && sm[nb] > kHD && heavy_degree(nb) > 1) rotors++;
    L.num_tors += 0.5f * rotors;
  }
  L.lines = r.lines;
  L.line_atom.assign(L.lines.size(), -1);
  for (int i = 0; i < n; i++) L.line_atom[b.atoms[i].line - 1] = i;
  const int nt = (int)b.parent.size() - 1;
  L.conf0.assign(7 + nt, 0.f);
  for (int d = 0; d < 3; d++) L.conf0[d] = b.origin[d];
  L.conf0[3] = 1.f;
  return L;
}

// Flexible receptor: rigid part + flexible residues (parse_receptor_pdbqt(rigid, flex), parse_pdbqt.cpp:419-527).
// A residue (BEGIN_RES ... END_RES) has the grammar of a ligand; postprocess_residue (:392-417) makes its ROOT atoms
// and the first atoms of its top-level branches `inflex` (fixed, but part of the model, not of the receptor grid)
// and every top-level branch a first_segment tree of movable atoms -- exactly node 0 / nodes >= 1 of the ligand
// construction above.  Bonds for the X-Score typing follow model::distance_type_between (model.cpp:491-508):
// rigid-rigid and rigid-inflex are fixed, rigid-movable is variable (never bonded), inside a residue the tree's
// own marks, inflex-inflex fixed, movable atoms of different residues variable.
namespace {
// The receptor model of parse_receptor_pdbqt(rigid, flex): grid_atoms = rigid, atoms = [movable | inflex]
// (pdbqt_initializer::initialize_from_rigid / initialize_from_nrp, parse_pdbqt.cpp:419-470), bonds and X-Score types
// assigned over the index space assign_bonds uses (grid atoms first, model.cpp:563-571).
struct FlexModel {
  struct Res {
    Builder b;
    std::vector<unsigned char> mobm;
    int n_inflex = 0;   // atoms of the Builder's node 0: ROOT atoms + first atoms of the top-level branches
    int mov_off = 0;    // index of the residue's first movable atom among all movable atoms
    int inflex_off = 0; // index of its first inflex atom among all inflex atoms
  };
  std::vector<Res> residues;
  struct Row {
    int res, idx;  // residue (-1 rigid), index in the residue's Builder order
    int kind;      // 0 movable, 1 inflex, 2 rigid
  };
  int n_rigid = 0, n_movable = 0, n_inflex = 0;
  std::vector<PAtom> atoms;       // [rigid | movable | inflex]
  std::vector<Row> rows;
  std::vector<std::vector<int>> bonds;
  std::vector<int> sm;
  int mob(int i, int j) const {   // model::distance_type_between (model.cpp:491-508)
    if (i == j) return 1;
    const Row &a = rows[i], &b = rows[j];
    if (a.kind == 2 || b.kind == 2) return (a.kind == 0 || b.kind == 0) ? 0 : 1;
    if (a.res == b.res) {
      const int n = (int)residues[a.res].b.atoms.size();
      return residues[a.res].mobm[(size_t)std::min(a.idx, b.idx) * n + std::max(a.idx, b.idx)];
    }
    return (a.kind == 1 && b.kind == 1) ? 1 : 0;
  }
};

FlexModel build_flex_model(const std::string &rigid_name, const std::string &rigid_text, const std::string &flex_name,
                           const std::string &flex_text) {
  FlexModel fm;
  {
    LineReader r(flex_name, flex_text);
    std::string s;
    while (r.next(s)) {  // parse_pdbqt_flex, :481-527
      if (s.empty() || starts_with(s, "WARNING") || starts_with(s, "REMARK") || starts_with(s, "USER")) continue;
      if (starts_with(s, "BEGIN_RES")) {
        PS root;
        (void)parse_tree(r, root, true);
        if (root.atoms.empty()) fail(flex_name, r.count, "No atoms in the residue");
        fm.residues.emplace_back();
        build_tree(root, fm.residues.back().b);
        fm.residues.back().mobm = mobility_of(fm.residues.back().b);
      } else if (starts_with(s, "MODEL")) {
        fail(flex_name, r.count, "Unexpected multi-MODEL input. Use \"vina_split\" first?");
      } else {
        fail(flex_name, r.count, "Unknown or inappropriate tag");
      }
    }
  }
  {
    LineReader r(rigid_name, rigid_text);
    std::string s;
    while (r.next(s)) {
      if (starts_with(s, "ATOM  ") || starts_with(s, "HETATM")) {
        fm.atoms.push_back(parse_atom(rigid_name, r.count, s));
        fm.rows.push_back({-1, 0, 2});
      } else if (starts_with(s, "MODEL")) {
        fail(rigid_name, r.count, "Unexpected multi-MODEL input. Use \"vina_split\" first?");
      }
    }
  }
  fm.n_rigid = (int)fm.atoms.size();
  for (int kind = 0; kind < 2; kind++)
    for (size_t q = 0; q < fm.residues.size(); q++) {
      FlexModel::Res &R = fm.residues[q];
      const Builder &b = R.b;
      (kind == 0 ? R.mov_off : R.inflex_off) = kind == 0 ? fm.n_movable : fm.n_inflex;
      for (int i = 0; i < (int)b.atoms.size(); i++) {
        const bool inflex = i >= b.abeg[0] && i < b.aend[0];
        if ((kind == 1) != inflex) continue;
        fm.atoms.push_back(b.atoms[i]);
        fm.rows.push_back({(int)q, i, kind});
        (kind == 0 ? fm.n_movable : fm.n_inflex)++;
        if (kind == 1) R.n_inflex++;
      }
    }
  bonds_and_types(fm.atoms, [&](int i, int j) { return fm.mob(i, j); }, fm.bonds, fm.sm);
  return fm;
}
}  // namespace

PdbqtFlexReceptor parse_pdbqt_receptor_flex(const std::string &rigid_name, const std::string &rigid_text,
                                            const std::string &flex_name, const std::string &flex_text) {
  const FlexModel fm = build_flex_model(rigid_name, rigid_text, flex_name, flex_text);
  // rows in DLScorer::setReceptor's order (dl_scorer.cpp:93-193): movable, inflex, rigid
  PdbqtFlexReceptor out;
  out.n_movable = fm.n_movable;
  out.n_inflex = fm.n_inflex;
  auto put = [&](int i) {
    out.xyz.insert(out.xyz.end(), fm.atoms[i].c, fm.atoms[i].c + 3);
    out.smt.push_back(fm.sm[i]);
  };
  for (int i = fm.n_rigid; i < (int)fm.atoms.size(); i++) put(i);
  for (int i = 0; i < fm.n_rigid; i++) put(i);
  return out;
}

// gnina's `model` for a docking run with flexible residues: parse_receptor_pdbqt(rigid, flex) + m.append(ligand)
// (molgetter.cpp:66-75,430-437; model::append, model.cpp:176-226).  Atoms = [flex movable | ligand | inflex]; nodes =
// [ligand root | ligand segments | one first_segment tree per top-level branch of every residue] so that torsion k
// belongs to node k + 1 and conf = [7 + T_ligand + T_flex] like gnina's (conf.h:361-373).
PdbqtModel parse_pdbqt_model(const std::string &rigid_name, const std::string &rigid_text, const std::string &flex_name,
                             const std::string &flex_text, const std::string &lig_name, const std::string &lig_text) {
  const FlexModel fm = build_flex_model(rigid_name, rigid_text, flex_name, flex_text);
  const PdbqtLigand lig = parse_pdbqt_ligand(lig_name, lig_text);
  PdbqtModel M;
  for (int i = 0; i < fm.n_rigid; i++) {
    M.rec_xyz.insert(M.rec_xyz.end(), fm.atoms[i].c, fm.atoms[i].c + 3);
    M.rec_smt.push_back(fm.sm[i]);
  }
  const int n_lig = (int)lig.smt.size(), n_mov = fm.n_movable, n_inf = fm.n_inflex;
  M.n_flex_movable = n_mov;
  M.n_inflex = n_inf;
  M.lig_begin = n_mov;
  M.lig_end = n_mov + n_lig;
  M.n_movable = n_mov + n_lig;
  M.torsdof = lig.torsdof;
  M.num_tors = lig.num_tors;
  const int n_atoms = n_mov + n_lig + n_inf;
  M.xyz.resize((size_t)3 * n_atoms);
  M.local_xyz.resize((size_t)3 * n_atoms);
  M.smt.resize(n_atoms);
  // receptor-model index (movable, then inflex) -> combined index (appender::operator(), model.cpp:84-98)
  auto rec_to_model = [&](int a) { return a < n_mov ? a : a + n_lig; };
  for (int a = 0; a < n_mov + n_inf; a++) {
    const int src = fm.n_rigid + a, dst = rec_to_model(a);
    for (int d = 0; d < 3; d++) M.xyz[3 * dst + d] = M.local_xyz[3 * dst + d] = fm.atoms[src].c[d];  // inflex: as in the file
    M.smt[dst] = fm.sm[src];
  }
  for (int i = 0; i < n_lig; i++) {
    for (int d = 0; d < 3; d++) {
      M.xyz[3 * (n_mov + i) + d] = lig.xyz[3 * i + d];
      M.local_xyz[3 * (n_mov + i) + d] = lig.local_xyz[3 * i + d];
    }
    M.smt[n_mov + i] = lig.smt[i];
  }
  // nodes: the ligand's first
  const int nl_nodes = (int)lig.node_parent.size();
  for (int k = 0; k < nl_nodes; k++) {
    M.node_parent.push_back(lig.node_parent[k]);
    M.node_atom_begin.push_back(lig.node_atom_begin[k] + n_mov);
    M.node_atom_end.push_back(lig.node_atom_end[k] + n_mov);
    for (int d = 0; d < 3; d++) {
      M.node_rel_origin.push_back(lig.node_rel_origin[3 * k + d]);
      M.node_rel_axis.push_back(lig.node_rel_axis[3 * k + d]);
    }
  }
  M.n_lig_torsions = nl_nodes - 1;
  // residues: Builder node k >= 1 -> model node; children of the Builder's node 0 are first_segments (world)
  for (const FlexModel::Res &R : fm.residues) {
    const Builder &b = R.b;
    const int base = (int)M.node_parent.size() - 1;  // Builder node k -> model node base + k
    for (int k = 1; k < (int)b.parent.size(); k++) {
      const bool first = b.parent[k] == 0;
      M.node_parent.push_back(first ? -2 : base + b.parent[k]);
      // the Builder numbers the residue's atoms [inflex (node 0) | movable]; movable atom i -> mov_off + i - n_inflex
      M.node_atom_begin.push_back(R.mov_off + b.abeg[k] - R.n_inflex);
      M.node_atom_end.push_back(R.mov_off + b.aend[k] - R.n_inflex);
      for (int d = 0; d < 3; d++) {
        M.node_rel_origin.push_back(first ? b.origin[3 * k + d] : b.rel_origin[3 * k + d]);
        M.node_rel_axis.push_back(b.rel_axis[3 * k + d]);  // built at the identity orientation: also the absolute axis
      }
    }
    // local coordinates of the residue's movable atoms (relative to their node's origin)
    for (int i = R.n_inflex; i < (int)b.atoms.size(); i++)
      for (int d = 0; d < 3; d++) M.local_xyz[3 * (R.mov_off + i - R.n_inflex) + d] = b.local[3 * i + d];
  }
  M.n_flex_torsions = (int)M.node_parent.size() - nl_nodes;
  // pairs.  (1) the receptor model's own other_pairs: initialize_pairs (model.cpp:682-703) over atoms = [movable |
  // inflex]; (2) model::append: every (receptor-model atom, ligand atom) pair, hydrogens included (model.cpp:182-199);
  // (3) the ligand's internal pairs.  other_pairs (kind 1) first: model::eval_deriv adds them first.
  const int nr = n_mov + n_inf;
  auto bonded_to3 = [&](int a) {  // model::bonded_to(a, 3): bonds to grid atoms are not followed (model.cpp:664-675)
    struct Rec {
      static void go(int a, int depth, const FlexModel &fm, std::vector<int> &out) {
        if (std::find(out.begin(), out.end(), a) != out.end()) return;
        out.push_back(a);
        if (depth > 0)
          for (int nb : fm.bonds[fm.n_rigid + a])
            if (nb >= fm.n_rigid) go(nb - fm.n_rigid, depth - 1, fm, out);
      }
    };
    std::vector<int> out;
    Rec::go(a, 3, fm, out);
    return out;
  };
  for (int i = 0; i < nr; i++) {
    const std::vector<int> near = bonded_to3(i);
    for (int j = i + 1; j < nr; j++) {
      if (i >= n_mov && j >= n_mov) continue;  // inflex-inflex
      if (fm.mob(fm.n_rigid + i, fm.n_rigid + j) != 0) continue;
      if (std::find(near.begin(), near.end(), j) != near.end()) continue;
      if (fm.sm[fm.n_rigid + i] <= kHD || fm.sm[fm.n_rigid + j] <= kHD) continue;
      M.pairs.push_back(rec_to_model(i));
      M.pairs.push_back(rec_to_model(j));
      M.pair_kind.push_back(1);
    }
  }
  for (int i = 0; i < nr; i++)
    for (int j = 0; j < n_lig; j++) {
      M.pairs.push_back(rec_to_model(i));
      M.pairs.push_back(n_mov + j);
      M.pair_kind.push_back(1);
    }
  for (size_t p = 0; p + 1 < lig.pairs.size(); p += 2) {
    M.pairs.push_back(lig.pairs[p] + n_mov);
    M.pairs.push_back(lig.pairs[p + 1] + n_mov);
    M.pair_kind.push_back(0);
  }
  M.conf0 = lig.conf0;
  M.conf0.resize(lig.conf0.size() + M.n_flex_torsions, 0.f);
  return M;
}

static std::string num9(float v) {  // boost::lexical_cast<std::string>(float): up to 9 significant digits
  char buf[48];
  snprintf(buf, sizeof buf, "%.9g", (double)v);
  return buf;
}

std::string write_pdbqt_pose(const PdbqtLigand &lig, const float *coords, int modelnum, float energy, float rmsd,
                             float cnnscore, float cnnaffinity) {
  std::ostringstream out;
  out << "MODEL " << modelnum << "\n";
  out << "REMARK minimizedAffinity " << num9(energy) << "\n";
  if (rmsd >= 0) out << "REMARK minimizedRMSD " << num9(rmsd) << "\n";
  if (cnnscore >= 0) out << "REMARK CNNscore " << num9(cnnscore) << "\n";
  if (cnnaffinity != 0) out << "REMARK CNNaffinity " << num9(cnnaffinity) << "\n";
  for (size_t i = 0; i < lig.lines.size(); i++) {
    std::string s = lig.lines[i];
    const int a = lig.line_atom[i];
    if (a >= 0) {  // coords_to_pdbqt_string: columns 31, 39, 47, width 8, 3 decimals
      for (int k = 0; k < 3; k++) {
        char buf[32];
        snprintf(buf, sizeof buf, "%8.3f", (double)coords[3 * a + k]);
        if (std::strlen(buf) != 8) throw std::runtime_error("coordinate does not fit the 8-column PDBQT field");
        s.replace(30 + 8 * k, 8, buf);
      }
    }
    out << s << "\n";
  }
  out << "ENDMDL\n";
  return out.str();
}

// One docked pose as gnina writes it to an .sdf (result_info::write's native branch, result_info.cpp:117-160, with the
// molecule block of sdfcontext::write, model.cpp:827-907): name, two blank lines, counts line, atom block (%10.4f x3,
// element), bond block, M  CHG / M  ISO properties, M  END, then the SD tags minimizedAffinity / minimizedRMSD /
// CNNscore / CNNaffinity + CNN_VS / CNNaffinity_variance, $$$$.  The connection table comes from the caller (gnina
// keeps the one OpenBabel read): elements [n_atoms][2] (not necessarily 0-terminated), atom_index [n_atoms] = model
// atom of every SDF atom, bonds [n_bonds][3] = (a, b, order), 0-based; props [n_props][3] = ('c' | 'i', atom, value).
std::string write_sdf_pose(const std::string &name, int n_atoms, const char *elements, const int32_t *atom_index,
                           const float *coords, int n_bonds, const int32_t *bonds, int n_props, const int32_t *props,
                           float energy, float rmsd, float cnnscore, float cnnaffinity, float cnnvariance) {
  std::ostringstream out;
  char buff[1024];
  out << name << "\n\n\n";
  snprintf(buff, sizeof buff, "%3d%3d  0  0  0  0  0  0  0  0999 V2000\n", n_atoms, n_bonds);
  out << buff;
  for (int i = 0; i < n_atoms; i++) {
    const float *c = coords + 3 * (atom_index ? atom_index[i] : i);
    char el[3] = {elements[2 * i], elements[2 * i + 1], 0};
    snprintf(buff, sizeof buff, "%10.4f%10.4f%10.4f %-3.2s 0  0  0  0  0  0  0  0  0  0  0  0\n", (double)c[0],
             (double)c[1], (double)c[2], el);
    out << buff;
  }
  for (int i = 0; i < n_bonds; i++)
    out << std::setw(3) << bonds[3 * i] + 1 << std::setw(3) << bonds[3 * i + 1] + 1 << std::setw(3) << bonds[3 * i + 2]
        << "  0\n";
  for (int i = 0; i < n_props; i++) {
    if (props[3 * i] == 'c')
      out << "M  CHG 1 " << std::setw(3) << props[3 * i + 1] + 1 << std::setw(4) << props[3 * i + 2] << "\n";
    else if (props[3 * i] == 'i')
      out << "M  ISO 1 " << std::setw(3) << props[3 * i + 1] + 1 << std::setw(4) << props[3 * i + 2] << "\n";
  }
  out << "M  END\n";
  out << "> <minimizedAffinity>\n" << std::fixed << std::setprecision(5) << energy << "\n\n";
  if (rmsd >= 0) out << "> <minimizedRMSD>\n" << std::fixed << std::setprecision(5) << rmsd << "\n\n";
  if (cnnscore >= 0) out << "> <CNNscore>\n" << std::fixed << std::setprecision(10) << cnnscore << "\n\n";
  if (cnnaffinity != 0) {
    out << "> <CNNaffinity>\n" << std::fixed << std::setprecision(10) << cnnaffinity << "\n\n";
    out << "> <CNN_VS>\n" << std::fixed << std::setprecision(10) << cnnaffinity * cnnscore << "\n\n";
  }
  if (cnnvariance != 0) out << "> <CNNaffinity_variance>\n" << std::fixed << std::setprecision(10) << cnnvariance << "\n\n";
  out << "$$$$\n";
  return out.str();
}

static std::string slurp(const std::string &path) {
  std::ifstream f(path, std::ios::binary);
  if (!f) throw std::runtime_error("could not open " + path);
  std::ostringstream ss;
  ss << f.rdbuf();
  return ss.str();
}

PdbqtReceptor read_pdbqt_receptor(const std::string &path) { return parse_pdbqt_receptor(path, slurp(path)); }
PdbqtFlexReceptor read_pdbqt_receptor_flex(const std::string &rigid_path, const std::string &flex_path) {
  return parse_pdbqt_receptor_flex(rigid_path, slurp(rigid_path), flex_path, slurp(flex_path));
}
PdbqtLigand read_pdbqt_ligand(const std::string &path) { return parse_pdbqt_ligand(path, slurp(path)); }
PdbqtModel read_pdbqt_model(const std::string &rigid_path, const std::string &flex_path, const std::string &lig_path) {
  return parse_pdbqt_model(rigid_path, slurp(rigid_path), flex_path, slurp(flex_path), lig_path, slurp(lig_path));
}

}  // namespace gnina_amd

// ---- C entry points ------------------------------------------------------------------------------------------
namespace {
thread_local std::string g_pdbqt_error;
}

struct mi_pdbqt_ligand {
  gnina_amd::PdbqtLigand L;
};
struct mi_pdbqt_model {
  gnina_amd::PdbqtModel M;
};

extern "C" {

const char *mi_pdbqt_last_error(void) { return g_pdbqt_error.c_str(); }

mi_status mi_pdbqt_read_receptor(const char *path, float *xyz, int32_t *smt, int capacity, int *n_atoms) {
  try {
    if (!path || !n_atoms) throw std::runtime_error("NULL argument");
    gnina_amd::PdbqtReceptor r = gnina_amd::read_pdbqt_receptor(path);
    *n_atoms = (int)r.smt.size();
    if (xyz && smt) {
      if ((int)r.smt.size() > capacity) throw std::runtime_error("capacity too small");
      if (!r.smt.empty()) {
        std::memcpy(xyz, r.xyz.data(), r.xyz.size() * sizeof(float));
        std::memcpy(smt, r.smt.data(), r.smt.size() * sizeof(int32_t));
      }
    }
    return MI_OK;
  } catch (const std::exception &e) {
    g_pdbqt_error = e.what();
    return MI_ERR_INVALID;
  }
}

mi_status mi_sdf_write_pose(const char *name, int n_atoms, const char *elements, const int32_t *atom_index,
                            const float *coords, int n_bonds, const int32_t *bonds, int n_props, const int32_t *props,
                            float energy, float rmsd, float cnnscore, float cnnaffinity, float cnnvariance, char *out,
                            size_t capacity, size_t *needed) {
  try {
    if (!name || !elements || !coords || !needed || n_atoms < 0 || n_bonds < 0 || n_props < 0 || (n_bonds && !bonds) ||
        (n_props && !props))
      throw std::runtime_error("bad arguments");
    if (n_atoms > 999 || n_bonds > 999) throw std::runtime_error("V2000 counts are limited to 999 atoms / bonds");
    const std::string s = gnina_amd::write_sdf_pose(name, n_atoms, elements, atom_index, coords, n_bonds, bonds, n_props,
                                                    props, energy, rmsd, cnnscore, cnnaffinity, cnnvariance);
    *needed = s.size() + 1;
    if (out) {
      if (capacity < s.size() + 1) throw std::runtime_error("capacity too small");
      std::memcpy(out, s.c_str(), s.size() + 1);
    }
    return MI_OK;
  } catch (const std::exception &e) {
    g_pdbqt_error = e.what();
    return MI_ERR_INVALID;
  }
}

mi_pdbqt_model *mi_pdbqt_model_open(const char *rigid, const char *flex, const char *ligand, int is_text) {
  try {
    if (!rigid || !flex || !ligand) throw std::runtime_error("NULL argument");
    std::unique_ptr<mi_pdbqt_model> h(new mi_pdbqt_model);
    h->M = is_text ? gnina_amd::parse_pdbqt_model("<rigid>", rigid, "<flex>", flex, "<ligand>", ligand)
                   : gnina_amd::read_pdbqt_model(rigid, flex, ligand);
    return h.release();
  } catch (const std::exception &e) {
    g_pdbqt_error = e.what();
    return nullptr;
  }
}
void mi_pdbqt_model_close(mi_pdbqt_model *h) { delete h; }
mi_status mi_pdbqt_model_sizes(const mi_pdbqt_model *h, int32_t *out8) {
  if (!h || !out8) return MI_ERR_INVALID;
  const gnina_amd::PdbqtModel &M = h->M;
  out8[0] = (int32_t)M.smt.size();
  out8[1] = (int32_t)M.node_parent.size();
  out8[2] = (int32_t)(M.pairs.size() / 2);
  out8[3] = (int32_t)M.rec_smt.size();
  out8[4] = M.n_flex_movable;
  out8[5] = M.n_inflex;
  out8[6] = M.n_lig_torsions;
  out8[7] = M.n_flex_torsions;
  return MI_OK;
}
mi_status mi_pdbqt_model_desc(const mi_pdbqt_model *h, mi_ligand_desc *desc, const float **xyz, const float **conf0,
                              const float **rec_xyz, const int32_t **rec_smt, float *num_tors) {
  if (!h || !desc) return MI_ERR_INVALID;
  const gnina_amd::PdbqtModel &M = h->M;
  *desc = mi_ligand_desc{};
  desc->n_atoms = (int32_t)M.smt.size();
  desc->smt = M.smt.data();
  desc->local_xyz = M.local_xyz.data();
  desc->n_nodes = (int32_t)M.node_parent.size();
  desc->node_parent = M.node_parent.data();
  desc->node_atom_begin = M.node_atom_begin.data();
  desc->node_atom_end = M.node_atom_end.data();
  desc->node_rel_origin = M.node_rel_origin.data();
  desc->node_rel_axis = M.node_rel_axis.data();
  desc->n_pairs = (int32_t)(M.pairs.size() / 2);
  desc->pairs = M.pairs.data();
  desc->n_movable = M.n_movable;
  desc->pair_kind = M.pair_kind.data();
  desc->lig_begin = M.lig_begin;
  desc->lig_end = M.lig_end;
  if (xyz) *xyz = M.xyz.data();
  if (conf0) *conf0 = M.conf0.data();
  if (rec_xyz) *rec_xyz = M.rec_xyz.data();
  if (rec_smt) *rec_smt = M.rec_smt.data();
  if (num_tors) *num_tors = M.num_tors;
  return MI_OK;
}

mi_status mi_pdbqt_read_receptor_flex(const char *rigid, const char *flex, int is_text, float *xyz, int32_t *smt,
                                      int capacity, int *n_atoms, int *n_movable, int *n_inflex) {
  try {
    if (!rigid || !flex || !n_atoms || !n_movable || !n_inflex) throw std::runtime_error("NULL argument");
    gnina_amd::PdbqtFlexReceptor r = is_text ? gnina_amd::parse_pdbqt_receptor_flex("<rigid>", rigid, "<flex>", flex)
                                             : gnina_amd::read_pdbqt_receptor_flex(rigid, flex);
    *n_atoms = (int)r.smt.size();
    *n_movable = r.n_movable;
    *n_inflex = r.n_inflex;
    if (xyz && smt) {
      if ((int)r.smt.size() > capacity) throw std::runtime_error("capacity too small");
      if (!r.smt.empty()) {
        std::memcpy(xyz, r.xyz.data(), r.xyz.size() * sizeof(float));
        std::memcpy(smt, r.smt.data(), r.smt.size() * sizeof(int32_t));
      }
    }
    return MI_OK;
  } catch (const std::exception &e) {
    g_pdbqt_error = e.what();
    return MI_ERR_INVALID;
  }
}

mi_pdbqt_ligand *mi_pdbqt_ligand_open(const char *path_or_text, int is_text) {
  try {
    if (!path_or_text) throw std::runtime_error("NULL argument");
    auto *h = new mi_pdbqt_ligand();
    h->L = is_text ? gnina_amd::parse_pdbqt_ligand("<text>", path_or_text) : gnina_amd::read_pdbqt_ligand(path_or_text);
    return h;
  } catch (const std::exception &e) {
    g_pdbqt_error = e.what();
    return nullptr;
  }
}

void mi_pdbqt_ligand_close(mi_pdbqt_ligand *h) { delete h; }

mi_status mi_pdbqt_ligand_num_tors(const mi_pdbqt_ligand *h, float *num_tors) {
  if (!h || !num_tors) return MI_ERR_INVALID;
  *num_tors = h->L.num_tors;
  return MI_OK;
}

mi_status mi_pdbqt_ligand_sizes(const mi_pdbqt_ligand *h, int *n_atoms, int *n_nodes, int *n_pairs, int *torsdof) {
  if (!h) return MI_ERR_INVALID;
  if (n_atoms) *n_atoms = (int)h->L.smt.size();
  if (n_nodes) *n_nodes = (int)h->L.node_parent.size();
  if (n_pairs) *n_pairs = (int)h->L.pairs.size() / 2;
  if (torsdof) *torsdof = h->L.torsdof;
  return MI_OK;
}

mi_status mi_pdbqt_write_pose(const mi_pdbqt_ligand *h, const float *coords, int modelnum, float energy, float rmsd,
                              float cnnscore, float cnnaffinity, char *out, size_t capacity, size_t *needed) {
  try {
    if (!h || !coords || !needed) throw std::runtime_error("NULL argument");
    const std::string s = gnina_amd::write_pdbqt_pose(h->L, coords, modelnum, energy, rmsd, cnnscore, cnnaffinity);
    *needed = s.size() + 1;
    if (out) {
      if (capacity < s.size() + 1) throw std::runtime_error("capacity too small");
      std::memcpy(out, s.c_str(), s.size() + 1);
    }
    return MI_OK;
  } catch (const std::exception &e) {
    g_pdbqt_error = e.what();
    return MI_ERR_INVALID;
  }
}

// The arrays of a mi_ligand_desc (pointers stay valid until mi_pdbqt_ligand_close), plus the input coordinates,
// the PDBQT serial numbers and the conformation that reproduces the input pose.
mi_status mi_pdbqt_ligand_desc(const mi_pdbqt_ligand *h, mi_ligand_desc *desc, const float **xyz, const int32_t **serial,
                               const float **conf0) {
  if (!h || !desc) return MI_ERR_INVALID;
  const gnina_amd::PdbqtLigand &L = h->L;
  desc->n_atoms = (int32_t)L.smt.size();
  desc->smt = L.smt.data();
  desc->local_xyz = L.local_xyz.data();
  desc->n_nodes = (int32_t)L.node_parent.size();
  desc->node_parent = L.node_parent.data();
  desc->node_atom_begin = L.node_atom_begin.data();
  desc->node_atom_end = L.node_atom_end.data();
  desc->node_rel_origin = L.node_rel_origin.data();
  desc->node_rel_axis = L.node_rel_axis.data();
  desc->n_pairs = (int32_t)(L.pairs.size() / 2);
  desc->pairs = L.pairs.data();
  if (xyz) *xyz = L.xyz.data();
  if (serial) *serial = L.serial.data();
  if (conf0) *conf0 = L.conf0.data();
  return MI_OK;
}
}
