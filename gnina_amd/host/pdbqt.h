// pdbqt.h -- native PDBQT reader for the hot path's inputs, no OpenBabel.
//
// Restates what gnina does between a .pdbqt file and the arrays the scoring path consumes:
//   * parse_pdbqt_rigid / parse_pdbqt_ligand_stream (gninasrc/lib/parse_pdbqt.cpp:145-183,185-305,419-435,
//     481-527): ATOM/HETATM columns, ROOT / BRANCH a b / ENDBRANCH / TORSDOF structure;
//   * postprocess_ligand / postprocess_branch (parse_pdbqt.cpp:346-391; parsing.h:122-212): atom order
//     (a branch's first atom -- its "immobile atom" -- is stored in the PARENT segment), segment frames
//     (tree.h:152-203: origin = immobile atom, axis = unit vector from the parent-side atom), the relative
//     mobility matrix (fixed / rotor / variable);
//   * model::initialize (model.cpp:560-720): covalent bonds from distances (assign_bonds), X-Score type
//     adjustment from bonded hydrogens / heteroatoms (adjust_smina_type, atom_constants.h:280-309), interacting
//     pairs (initialize_pairs: variable distance, more than 3 bonds apart, heavy atoms).
// The output is exactly what mi_scorer_set_receptor / mi_vina_set_receptor and mi_vina_set_ligand take.
//
// Flexible receptors: parse_receptor_pdbqt(rigid, flex) -- residues between BEGIN_RES / END_RES have the grammar of a
// ligand (parse_pdbqt.cpp:392-417,419-527); the reader returns the atoms in DLScorer::setReceptor's row order
// (movable, inflex, rigid) with the X-Score types of the combined model (the residues' torsion trees are parsed
// and used for the bond mobility, not yet returned).
// Not restated: the `fix_hydrogens` option (off by default in gnina), multi-MODEL files.  Parity: "unpinned" -- the reference ships no ligand .pdbqt fixture with expected types or pairs, so
// the tests check hand-derived small molecules and structural invariants (tests/test_pdbqt_cpu.py).
#pragma once
#include <cstdint>
#include <string>
#include <vector>

namespace gnina_amd {

struct PdbqtReceptor {
  std::vector<float> xyz;     // [n][3]
  std::vector<int32_t> smt;   // [n] smina types after adjust_smina_type
};

struct PdbqtLigand {
  // atoms in model order (node by node, DFS pre-order of the torsion tree)
  std::vector<float> xyz;        // [n][3] input coordinates
  std::vector<int32_t> smt;      // [n]
  std::vector<int32_t> serial;   // [n] PDBQT atom numbers
  std::vector<float> local_xyz;  // [n][3] relative to the owning node's origin
  // nodes: 0 = rigid root, k > 0 = segment of torsion k-1
  std::vector<int32_t> node_parent, node_atom_begin, node_atom_end;
  std::vector<float> node_rel_origin, node_rel_axis;  // [n_nodes][3]
  std::vector<int32_t> pairs;    // [n_pairs][2], a < b
  std::vector<float> conf0;      // [7 + T]: root origin, identity quaternion, zero torsions = the input pose
  int torsdof = 0;               // TORSDOF record
  float num_tors = 0;            // conf_independent_inputs::num_tors (terms.cpp:74-106), what num_tors_div divides by
  // the file's lines and, for ATOM/HETATM lines, the model index of the atom (-1 otherwise): gnina's `context`
  // (model.h:205-230), used to write poses back in the input's own format
  std::vector<std::string> lines;
  std::vector<int32_t> line_atom;
};

struct PdbqtFlexReceptor {
  // rows: [0, n_movable) movable atoms of the flexible residues (residue by residue, tree order),
  // [n_movable, n_movable + n_inflex) their fixed atoms (ROOT atoms and the first atom of every top-level branch),
  // then the rigid receptor -- the order of DLScorer::setReceptor (dl_scorer.cpp:93-193): the first n_movable rows
  // are what mi_scorer_set_flex declares
  std::vector<float> xyz;    // [n][3]
  std::vector<int32_t> smt;  // [n] after adjust_smina_type on the combined model
  int n_movable = 0, n_inflex = 0;
};
PdbqtFlexReceptor read_pdbqt_receptor_flex(const std::string &rigid_path, const std::string &flex_path);
PdbqtFlexReceptor parse_pdbqt_receptor_flex(const std::string &rigid_name, const std::string &rigid_text,
                                            const std::string &flex_name, const std::string &flex_text);

// gnina's `model` of a docking run with flexible residues: rigid receptor (the grid atoms) + what moves.
// Atoms = [flexible-residue movable atoms | ligand | inflex atoms of the residues] (model.h: movable atoms first),
// nodes = [ligand root | ligand segments | the residues' trees], conf = [7 + T_ligand + T_flex].
struct PdbqtModel {
  std::vector<float> rec_xyz;      // [n_rigid][3] receptor atoms for the grids (mi_vina_set_receptor)
  std::vector<int32_t> rec_smt;    // typed within the receptor model (rigid + residues), like the reference
  std::vector<float> xyz;          // [n_atoms][3] input coordinates in model order
  std::vector<int32_t> smt;
  std::vector<float> local_xyz;    // [n_atoms][3] relative to the owning node's origin; inflex atoms: absolute
  std::vector<int32_t> node_parent;  // -1 ligand root, -2 a residue's first segment (hangs off the world), else index
  std::vector<int32_t> node_atom_begin, node_atom_end;
  std::vector<float> node_rel_origin, node_rel_axis;  // first segments: absolute origin / axis
  std::vector<int32_t> pairs;      // [n_pairs][2]
  std::vector<int32_t> pair_kind;  // [n_pairs] 1 = model::other_pairs (cap v[2]), 0 = ligand-internal (cap v[0])
  std::vector<float> conf0;        // [7 + T_ligand + T_flex]: the input pose
  int n_movable = 0, n_flex_movable = 0, n_inflex = 0, lig_begin = 0, lig_end = 0;
  int n_lig_torsions = 0, n_flex_torsions = 0, torsdof = 0;
  float num_tors = 0;
};
PdbqtModel read_pdbqt_model(const std::string &rigid_path, const std::string &flex_path, const std::string &lig_path);
PdbqtModel parse_pdbqt_model(const std::string &rigid_name, const std::string &rigid_text, const std::string &flex_name,
                             const std::string &flex_text, const std::string &lig_name, const std::string &lig_text);

// One docked pose as gnina writes it to a .pdbqt (result_info::write, result_info.cpp:151-164 +
// context::writePDBQT / coords_to_pdbqt_string, model.cpp:779-810): MODEL n, REMARK minimizedAffinity /
// [minimizedRMSD] / [CNNscore] / [CNNaffinity], the input lines with columns 31-54 rewritten (%8.3f), ENDMDL.
// coords [n_atoms][3] in model order; rmsd < 0 and cnnscore < 0 omit their remarks, cnnaffinity == 0 omits its.
std::string write_pdbqt_pose(const PdbqtLigand &lig, const float *coords, int modelnum, float energy, float rmsd,
                             float cnnscore, float cnnaffinity);

// One docked pose in gnina's .sdf output (result_info::write native branch + sdfcontext::write): see pdbqt.cpp.
std::string write_sdf_pose(const std::string &name, int n_atoms, const char *elements, const int32_t *atom_index,
                           const float *coords, int n_bonds, const int32_t *bonds, int n_props, const int32_t *props,
                           float energy, float rmsd, float cnnscore, float cnnaffinity, float cnnvariance);

// Throw std::runtime_error("<name>:<line>: <what>") on malformed input, like parse_error.
PdbqtReceptor read_pdbqt_receptor(const std::string &path);
PdbqtLigand read_pdbqt_ligand(const std::string &path);
PdbqtReceptor parse_pdbqt_receptor(const std::string &name, const std::string &text);
PdbqtLigand parse_pdbqt_ligand(const std::string &name, const std::string &text);

}  // namespace gnina_amd
