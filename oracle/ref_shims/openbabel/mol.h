// Stand-in for the OpenBabel classes the reference's Vina sources name (oracle/_ref only).  OpenBabel does
// molecule perception and file conversion for gnina's front end; the Vina path under test starts from PDBQT text
// (parse_pdbqt.cpp), so these are empty molecules: every method the reference calls exists, none does chemistry.
#pragma once
#include <iostream>
#include <string>
#include <vector>
namespace OpenBabel {
class OBMol;
class OBAtom {
 public:
  unsigned GetAtomicNum() const { return 0; }
  bool IsAromatic() const { return false; }
  bool IsHbondAcceptor() const { return false; }
  double GetX() const { return 0; }
  double GetY() const { return 0; }
  double GetZ() const { return 0; }
  double x() const { return 0; }
  double y() const { return 0; }
  double z() const { return 0; }
  unsigned GetIdx() const { return 0; }
  std::vector<OBAtom *> nbrs;
};
class OBBond {};
class OBMol {
 public:
  std::vector<OBAtom *> atoms;
  unsigned NumAtoms() const { return 0; }
  OBMol &operator+=(const OBMol &) { return *this; }
  void SetChainsPerceived(bool = true) {}
  void ConnectTheDots() {}
  void Clear() {}
  const char *GetTitle() const { return ""; }
};
namespace OBElements {
inline const char *GetSymbol(unsigned) { return "Xx"; }
}
class OBConversion {
 public:
  enum Option_type { INOPTIONS, OUTOPTIONS, GENOPTIONS };
  bool SetOutFormat(const char *) { return false; }
  bool SetInFormat(const char *) { return false; }
  void AddOption(const char *, Option_type, const char * = nullptr) {}
  bool ReadString(OBMol *, const std::string &) { return false; }
  bool Write(OBMol *, std::ostream * = nullptr) { return false; }
  bool Read(OBMol *, std::istream * = nullptr) { return false; }
};
}  // namespace OpenBabel
#define FOR_NBORS_OF_ATOM(n, a) for (OpenBabel::OBAtom * n : (a).nbrs)
#define FOR_ATOMS_OF_MOL(a, m) for (OpenBabel::OBAtom * a : (m).atoms)
