#include "model.h"

#include <cmath>
#include <cstring>
#include <sstream>

namespace mig {

int ModelDesc::grid_points() const {
  // N = round(dimension / resolution) + 1  (gninasrc/gninagrid/molgridder.cpp:48)
  return (int)std::lround((double)dimension / (double)resolution) + 1;
}

static std::vector<std::string> split_ws(const std::string &s) {
  std::vector<std::string> out;
  std::istringstream is(s);
  std::string t;
  while (is >> t) out.push_back(t);
  return out;
}

ModelDesc parse_blob(const void *blob, size_t nbytes, const char *name_override) {
  const unsigned char *p = (const unsigned char *)blob;
  MIG_CHECK(blob && nbytes >= 12 && std::memcmp(p, "MIGNINA1", 8) == 0, 2, "not a MIGNINA1 weight blob");
  uint32_t hl;
  std::memcpy(&hl, p + 8, 4);
  MIG_CHECK((size_t)12 + hl <= nbytes, 2, "truncated MIGNINA1 header");
  std::string header((const char *)p + 12, hl);
  size_t off = 12 + (size_t)hl;
  off += (64 - off % 64) % 64;

  ModelDesc m;
  std::vector<std::vector<std::string>> rec_lines, lig_lines;
  long ndata = -1;
  std::istringstream hs(header);
  std::string line;
  try {
    while (std::getline(hs, line)) {
      auto t = split_ws(line);
      if (t.empty()) continue;
      const std::string &k = t[0];
      if (k == "name") m.name = t.at(1);
      else if (k == "family") m.family = t.at(1);
      else if (k == "resolution") m.resolution = std::stof(t.at(1));
      else if (k == "dimension") m.dimension = std::stof(t.at(1));
      else if (k == "radius_scaling") m.radius_scaling = std::stof(t.at(1));
      else if (k == "skip_softmax") m.skip_softmax = std::stoi(t.at(1)) != 0;
      else if (k == "apply_logistic_loss") m.apply_logistic_loss = std::stoi(t.at(1)) != 0;
      else if (k == "recmap") rec_lines.emplace_back(t.begin() + 1, t.end());
      else if (k == "ligmap") lig_lines.emplace_back(t.begin() + 1, t.end());
      else if (k == "ndata") ndata = std::stol(t.at(1));
      else if (k == "buf") {
        int id = std::stoi(t.at(1));
        MIG_CHECK(id == (int)m.bufs.size(), 2, "buffer ids must be sequential");
        BufDecl b;
        b.S = std::stoi(t.at(2));
        b.C = std::stoi(t.at(3));
        m.bufs.push_back(b);
      } else if (k == "pool") {
        Op o;
        o.kind = OpKind::Pool;
        o.pool_mode = t.at(1) == "max" ? 1 : (t.at(1) == "avg" ? 2 : 0);
        MIG_CHECK(o.pool_mode != 0, 2, "bad pool mode");
        o.src = std::stoi(t.at(2));
        o.dst = std::stoi(t.at(3));
        m.ops.push_back(o);
      } else if (k == "conv") {
        Op o;
        o.kind = OpKind::Conv;
        o.ksize = std::stoi(t.at(1));
        o.src = std::stoi(t.at(2));
        o.dst = std::stoi(t.at(3));
        o.cin = std::stoi(t.at(4));
        o.cout = std::stoi(t.at(5));
        o.dst_c0 = std::stoi(t.at(6));
        o.relu = std::stoi(t.at(7));
        o.w_off = std::stol(t.at(8));
        o.b_off = std::stol(t.at(9));
        o.bn_scale_off = std::stol(t.at(10));
        o.bn_shift_off = std::stol(t.at(11));
        MIG_CHECK(o.ksize == 1 || o.ksize == 3, 2, "conv kernel size must be 1 or 3");
        m.ops.push_back(o);
      } else if (k == "gmax") {
        Op o;
        o.kind = OpKind::GMax;
        o.src = std::stoi(t.at(1));
        o.dst = std::stoi(t.at(2));
        m.ops.push_back(o);
      } else if (k == "overlap") {  // test/gnina/data/overlap*.pt: mean over the grid of rec * lig density
        Op o;
        o.kind = OpKind::Overlap;
        o.src = o.dst = std::stoi(t.at(1));
        m.ops.push_back(o);
      } else if (k == "fc") {
        Op o;
        o.kind = OpKind::Fc;
        o.src = std::stoi(t.at(1));
        o.n_in = std::stoi(t.at(2));
        o.w_off = std::stol(t.at(3));
        o.b_off = std::stol(t.at(4));
        m.ops.push_back(o);
      }
    }
    m.recmap.build(rec_lines);
    m.ligmap.build(lig_lines);
  } catch (const Error &) {
    throw;
  } catch (const std::string &s) {
    throw Error(2, s);
  } catch (const std::exception &e) {
    throw Error(2, std::string("malformed MIGNINA1 header: ") + e.what());
  }
  MIG_CHECK(ndata >= 0 && off <= nbytes && (size_t)ndata <= (nbytes - off) / 4, 2, "truncated MIGNINA1 payload");
  MIG_CHECK(m.resolution > 0 && m.dimension > 0 && std::isfinite(m.resolution) && std::isfinite(m.dimension), 2,
            "resolution and dimension must be positive");
  m.data.resize((size_t)ndata);
  std::memcpy(m.data.data(), p + off, (size_t)ndata * 4);
  if (name_override && *name_override) m.name = name_override;

  // structural validation of the program
  MIG_CHECK(!m.bufs.empty() && !m.ops.empty(), 2, "empty layer program");
  for (const BufDecl &b : m.bufs) MIG_CHECK(b.S > 0 && b.C > 0, 2, "buffer sizes must be positive");
  MIG_CHECK(m.bufs[0].S == m.grid_points() && m.bufs[0].C == m.n_channels(), 2,
            "buffer 0 must be the voxel grid [N][N][N][n_rec+n_lig channels]");
  auto okbuf = [&](int b) { return b >= 0 && b < (int)m.bufs.size(); };
  for (const Op &o : m.ops) {
    if (o.kind != OpKind::Fc) MIG_CHECK(okbuf(o.src) && okbuf(o.dst), 2, "op references unknown buffer");
    if (o.kind == OpKind::Conv) {
      MIG_CHECK(o.cin > 0 && o.cout > 0 && o.dst_c0 >= 0, 2, "conv channel counts must be positive");
      MIG_CHECK(o.cin <= m.bufs[o.src].C && o.dst_c0 + o.cout <= m.bufs[o.dst].C, 2, "conv channel range");
      MIG_CHECK(m.bufs[o.src].S == m.bufs[o.dst].S, 2, "conv changes spatial size");
      long nw = (long)o.ksize * o.ksize * o.ksize * o.cin * o.cout;
      MIG_CHECK(o.w_off >= 0 && o.w_off + nw <= ndata && o.b_off >= 0 && o.b_off + o.cout <= ndata, 2,
                "conv weights out of range");
      MIG_CHECK(o.bn_scale_off >= 0 || o.bn_shift_off < 0, 2, "bn shift without bn scale");
      if (o.bn_scale_off >= 0)
        MIG_CHECK(o.bn_scale_off + o.cin <= ndata && o.bn_shift_off >= 0 && o.bn_shift_off + o.cin <= ndata, 2,
                  "bn params out of range");
    } else if (o.kind == OpKind::Pool) {
      MIG_CHECK(m.bufs[o.src].S == 2 * m.bufs[o.dst].S && m.bufs[o.src].C <= m.bufs[o.dst].C, 2, "pool shapes");
    } else if (o.kind == OpKind::GMax) {
      MIG_CHECK(m.bufs[o.dst].S == 1 && m.bufs[o.src].C <= m.bufs[o.dst].C, 2, "gmax shapes");
    } else if (o.kind == OpKind::Fc) {
      MIG_CHECK(okbuf(o.src), 2, "fc references unknown buffer");
      const BufDecl &b = m.bufs[o.src];
      MIG_CHECK(o.n_in == b.S * b.S * b.S * b.C, 2, "fc input size");
      MIG_CHECK(o.w_off >= 0 && o.w_off + 3L * o.n_in <= ndata && o.b_off >= 0 && o.b_off + 3 <= ndata, 2,
                "fc weights out of range");
    }
  }
  if (m.ops.size() == 1 && m.ops[0].kind == OpKind::Overlap) {
    MIG_CHECK(m.ops[0].src == 0 && m.bufs[0].C == 2, 2, "the overlap model takes the two-channel voxel grid");
    return m;
  }
  MIG_CHECK(m.ops.back().kind == OpKind::Fc, 2, "program must end with the fc heads");
  MIG_CHECK(m.ops.front().kind == OpKind::Pool && m.ops.front().src == 0, 2,
            "program must start by pooling the voxel grid (all shipped families do)");
  return m;
}

// Same network on a different grid (metadata "resolution" / "dimension" of a user-supplied model file,
// torch_model.cpp:73-84): only networks whose head does not depend on the spatial size qualify -- the
// dynamic-global-pool Dense family; Default2017/2018 heads are fixed at 128 * 6^3 inputs and
// `dense.pt` hard-codes max_pool3d(6).
void regrid(ModelDesc &m, float resolution, float dimension) {
  MIG_CHECK(resolution > 0 && dimension > 0, 2, "resolution and dimension must be positive");
  const int old_n = m.grid_points();
  m.resolution = resolution;
  m.dimension = dimension;
  const int new_n = m.grid_points();
  if (new_n == old_n) return;
  MIG_CHECK(m.family == "Dense", 2,
            "model " + m.name + " (" + m.family + ") has a fixed-size head and cannot run on a " +
                std::to_string(new_n) + "^3 grid");
  MIG_CHECK(new_n % 8 == 0, 2, "grid points per side must be a multiple of 8 (three 2x2x2 pools)");
  for (const Op &o : m.ops)
    MIG_CHECK(o.kind != OpKind::Fc || m.bufs[o.src].S == 1, 2, "fc head depends on the spatial size");
  for (BufDecl &b : m.bufs) {
    if (b.S == 1) continue;  // global pool output
    MIG_CHECK(old_n % b.S == 0, 2, "unexpected buffer size");
    const int div = old_n / b.S;
    MIG_CHECK(new_n % div == 0, 2, "grid does not divide through the pooling stack");
    b.S = new_n / div;
  }
}

}  // namespace mig
