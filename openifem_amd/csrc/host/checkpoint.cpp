// checkpoint.cpp -- FluidSolver::save_checkpoint / load_checkpoint of the host mirror
// (source/mpi_fluid_solver.cpp:582-713).  Same protocol as the reference: files are named after the time step
// (<6 digits>.fluid_checkpoint, + .fluid_checkpoint.info and .fluid_checkpoint_fixed.data), only the latest older
// checkpoint is kept when a new one is written, load_checkpoint() picks the newest one in the directory, rebuilds the
// system, restores present_solution and replays Time (and the .pvd collection, and the time of hard-coded boundary
// Fields) up to the saved step.  The payload is this build's own: deal.II serialises a p4est forest, which has no
// meaning without deal.II; here every rank stores its owned node values keyed by global node id, so a checkpoint can be
// restored on a different number of GPUs (box partitions; single-rank numberings are restored on a single rank).
#include <algorithm>
#include <cstdio>
#include <cstring>
#include <filesystem>
#include <fstream>
#include <iomanip>
#include <set>
#include <sstream>
#include <unordered_map>
#include "insim.hpp"

namespace fs = std::filesystem;

namespace ifem_host {
namespace Fluid {
namespace MPI {

namespace {
constexpr char kMagic[] = "OpenIFEM-HIP fluid checkpoint 1";

std::string six(unsigned v) {
  std::ostringstream s;
  s << std::setw(6) << std::setfill('0') << v;
  return s.str();
}
std::string data_name(const std::string &stem, int rank, int nranks) {
  return stem + ".fluid_checkpoint_fixed.data" + (nranks > 1 ? "." + std::to_string(rank) : std::string());
}
struct Header {
  int dim = 0, kv = 0, nranks = 0, keyed_global = 0;
  long long n_unodes_global = 0, n_pnodes_global = 0;
  unsigned timestep = 0;
  double time = 0;
};
template <class T>
void put(std::ofstream &o, const T *p, size_t n) { o.write(reinterpret_cast<const char *>(p), std::streamsize(n * sizeof(T))); }
template <class T>
void get(std::ifstream &i, T *p, size_t n) {
  i.read(reinterpret_cast<char *>(p), std::streamsize(n * sizeof(T)));
  if (!i) throw std::runtime_error("load_checkpoint: truncated data file");
}
} // namespace

template <int dim>
void FluidSolver<dim>::save_checkpoint(const int output_index) {
  const int nranks = proc_grid[0] * proc_grid[1] * proc_grid[2];
  const fs::path dir(output_dir.empty() ? std::string("./") : output_dir);
  if (part_rank == 0) { // keep only the latest earlier checkpoint (:584-614)
    std::set<fs::path> checkpoints;
    for (const auto &p : fs::directory_iterator(dir))
      if (p.path().extension() == ".fluid_checkpoint") checkpoints.insert(p.path());
    while (checkpoints.size() > 1) {
      if (pcout) *pcout << "Removing " << *checkpoints.begin() << std::endl;
      const fs::path old = *checkpoints.begin();
      const std::string stem = (old.parent_path() / old.stem()).string();
      std::error_code ec;
      fs::remove(old, ec);
      fs::remove(stem + ".fluid_checkpoint.info", ec);
      // every piece of the old checkpoint, whatever rank count wrote it: <stem>.fluid_checkpoint_fixed.data[.<rank>]
      const std::string piece = old.stem().string() + ".fluid_checkpoint_fixed.data";
      std::vector<fs::path> pieces;
      for (const auto &q : fs::directory_iterator(dir))
        if (q.path().filename().string().compare(0, piece.size(), piece) == 0) pieces.push_back(q.path());
      for (const auto &q : pieces) fs::remove(q, ec);
      checkpoints.erase(checkpoints.begin());
    }
  }
  const std::string stem = (dir / six((unsigned)output_index)).string();
  const bool keyed = (int64_t)part.l2g_u.size() >= dofs.n_unodes_owned && (int64_t)part.l2g_p.size() >= dofs.n_pnodes_owned && dofs.n_unodes_owned > 0;
  if (nranks > 1 && !keyed) throw std::runtime_error("save_checkpoint: a partitioned run needs global node ids");
  const int64_t nuo = dofs.n_unodes_owned, npo = dofs.n_pnodes_owned;
  const std::vector<double> sol = get_current_solution(); // [u local (owned first) | p local]
  {
    std::ofstream o(data_name(stem, part_rank, nranks), std::ios::binary);
    if (!o) throw std::runtime_error("save_checkpoint: cannot open " + data_name(stem, part_rank, nranks));
    const int64_t counts[2] = {nuo, npo};
    put(o, counts, 2);
    std::vector<int64_t> gid((size_t)std::max(nuo, npo));
    for (int64_t i = 0; i < nuo; ++i) gid[(size_t)i] = keyed ? part.l2g_u[(size_t)i] : i;
    put(o, gid.data(), (size_t)nuo);
    put(o, sol.data(), (size_t)nuo * dim);
    for (int64_t i = 0; i < npo; ++i) gid[(size_t)i] = keyed ? part.l2g_p[(size_t)i] : i;
    put(o, gid.data(), (size_t)npo);
    put(o, sol.data() + dofs.n_u(), (size_t)npo);
  }
  {
    double all_written = 0; // collective: every rank's piece is on disk before the marker appears
    check(ifem_vec_norm2(ctx, IFEM_VEC_PRESENT, &all_written), "save_checkpoint");
  }
  if (part_rank == 0) {
    const long long ngu = part.n_unodes_global, ngp = part.n_pnodes_global;
    std::ofstream info(stem + ".fluid_checkpoint.info");
    info << "version dim velocity_degree ranks keyed_by_global_id n_velocity_nodes n_pressure_nodes timestep time\n"
         << 1 << ' ' << dim << ' ' << dofs.kv << ' ' << nranks << ' ' << int(keyed) << ' ' << ngu << ' ' << ngp << ' '
         << time.get_timestep() << ' ' << std::setprecision(17) << time.current() << '\n';
    // the marker file is written last: a checkpoint without it is ignored by load_checkpoint
    std::ofstream o(stem + ".fluid_checkpoint");
    o << kMagic << '\n';
  }
  if (pcout) *pcout << "Checkpoint file successfully saved at time step " << output_index << "!" << std::endl;
}

template <int dim>
bool FluidSolver<dim>::load_checkpoint() {
  const int nranks = proc_grid[0] * proc_grid[1] * proc_grid[2];
  const fs::path dir(output_dir.empty() ? std::string("./") : output_dir);
  fs::path checkpoint_file; // the latest one: largest stem (:648-658)
  if (fs::exists(dir))
    for (const auto &p : fs::directory_iterator(dir))
      if (p.path().extension() == ".fluid_checkpoint" &&
          (checkpoint_file.empty() || p.path().stem().string() > checkpoint_file.stem().string()))
        checkpoint_file = p.path();
  if (checkpoint_file.empty()) {
    if (pcout) *pcout << "Did not find fluid checkpoint files. Start from the beginning !" << std::endl;
    return false;
  }
  if (pcout) *pcout << "Loading checkpoint file " << checkpoint_file.filename().string() << "!" << std::endl;
  const std::string stem = (checkpoint_file.parent_path() / checkpoint_file.stem()).string();
  Header h;
  {
    std::ifstream info(stem + ".fluid_checkpoint.info");
    std::string line;
    int version = 0;
    if (!info || !std::getline(info, line) ||
        !(info >> version >> h.dim >> h.kv >> h.nranks >> h.keyed_global >> h.n_unodes_global >> h.n_pnodes_global >> h.timestep >> h.time) ||
        version != 1)
      throw std::runtime_error("load_checkpoint: unreadable " + stem + ".fluid_checkpoint.info");
  }
  // the mesh is regenerated, not stored: same generator, same refinement as a fresh start
  triangulation.refine_global(parameters.global_refinements[0]);
  setup_dofs();
  make_constraints();
  initialize_system();
  const bool keyed = (int64_t)part.l2g_u.size() >= dofs.n_unodes_owned && (int64_t)part.l2g_p.size() >= dofs.n_pnodes_owned && dofs.n_unodes_owned > 0;
  const long long ngu = part.n_unodes_global, ngp = part.n_pnodes_global;
  if (h.dim != dim || h.kv != dofs.kv || h.n_unodes_global != ngu || h.n_pnodes_global != ngp)
    throw std::runtime_error("load_checkpoint: the checkpoint belongs to a different mesh or element");
  const int64_t nuo = dofs.n_unodes_owned, npo = dofs.n_pnodes_owned;
  std::vector<double> sol((size_t)dofs.n_dofs(), 0.0);
  std::vector<uint8_t> seen_u((size_t)nuo, 0), seen_p((size_t)npo, 0);
  auto read_piece = [&](int r, bool same_layout, const std::unordered_map<int64_t, int64_t> *mu, const std::unordered_map<int64_t, int64_t> *mp) {
    std::ifstream in(data_name(stem, r, h.nranks), std::ios::binary);
    if (!in) throw std::runtime_error("load_checkpoint: missing " + data_name(stem, r, h.nranks));
    int64_t counts[2];
    get(in, counts, 2);
    // validate the header before anything is sized or indexed by it (a stale or corrupt piece must not write out of bounds)
    if (same_layout && (counts[0] != nuo || counts[1] != npo))
      throw std::runtime_error("load_checkpoint: piece size differs from this rank's owned range");
    if (counts[0] < 0 || counts[1] < 0 || counts[0] > h.n_unodes_global || counts[1] > h.n_pnodes_global)
      throw std::runtime_error("load_checkpoint: corrupt piece header in " + data_name(stem, r, h.nranks));
    std::vector<int64_t> gid((size_t)counts[0]);
    std::vector<double> val((size_t)counts[0] * dim);
    get(in, gid.data(), gid.size());
    get(in, val.data(), val.size());
    for (int64_t i = 0; i < counts[0]; ++i) {
      int64_t l = -1;
      if (same_layout) l = i;
      else { auto it = mu->find(gid[(size_t)i]); if (it != mu->end()) l = it->second; }
      if (l < 0) continue;
      if (same_layout && (keyed ? part.l2g_u[(size_t)l] : l) != gid[(size_t)i])
        throw std::runtime_error("load_checkpoint: node numbering of the checkpoint differs from this run");
      for (int c = 0; c < dim; ++c) sol[(size_t)l * dim + c] = val[(size_t)i * dim + c];
      seen_u[(size_t)l] = 1;
    }
    gid.resize((size_t)counts[1]);
    val.resize((size_t)counts[1]);
    get(in, gid.data(), gid.size());
    get(in, val.data(), val.size());
    for (int64_t i = 0; i < counts[1]; ++i) {
      int64_t l = -1;
      if (same_layout) l = i;
      else { auto it = mp->find(gid[(size_t)i]); if (it != mp->end()) l = it->second; }
      if (l < 0) continue;
      if (same_layout && (keyed ? part.l2g_p[(size_t)l] : l) != gid[(size_t)i])
        throw std::runtime_error("load_checkpoint: node numbering of the checkpoint differs from this run");
      sol[(size_t)dofs.n_u() + (size_t)l] = val[(size_t)i];
      seen_p[(size_t)l] = 1;
    }
  };
  if (h.nranks == nranks) // same partition: this rank's own piece, position by position
    read_piece(part_rank, true, nullptr, nullptr);
  else {
    if (!h.keyed_global || !keyed)
      throw std::runtime_error("load_checkpoint: restoring on a different number of ranks needs global node ids on both sides");
    std::unordered_map<int64_t, int64_t> mu, mp;
    mu.reserve((size_t)nuo * 2); mp.reserve((size_t)npo * 2);
    for (int64_t i = 0; i < nuo; ++i) mu[part.l2g_u[(size_t)i]] = i;
    for (int64_t i = 0; i < npo; ++i) mp[part.l2g_p[(size_t)i]] = i;
    for (int r = 0; r < h.nranks; ++r) read_piece(r, false, &mu, &mp);
  }
  if (std::find(seen_u.begin(), seen_u.end(), 0) != seen_u.end() || std::find(seen_p.begin(), seen_p.end(), 0) != seen_p.end())
    throw std::runtime_error("load_checkpoint: the checkpoint does not cover every owned node");
  check(ifem_vec_set(ctx, IFEM_VEC_PRESENT, sol.data()), "load_checkpoint");
  check(ifem_halo_exchange(ctx, IFEM_VEC_PRESENT), "load_checkpoint"); // ghosted present_solution = tmp (:679)
  // set the current time and write the .pvd records of the steps already done (:692-711)
  const unsigned target = (unsigned)std::stoul(checkpoint_file.stem().string()); // the step is the file name, as in the reference
  for (unsigned i = 0; i <= target; ++i) {
    if (output_enabled && (time.current() == 0 || time.time_to_output()) && part_rank == 0) {
      if (!pvd_writer) pvd_writer.reset(new Utils::PVDWriter(time, output_dir + "fluid.pvd"));
      pvd_writer->write_current_timestep("fluid_", 6);
    }
    if (i == target) break;
    time.increment();
    if (!hard_coded_boundary_values.empty()) field_time += time.get_delta_t();
  }
  if (pcout) *pcout << "Checkpoint file successfully loaded from time step " << time.get_timestep() << "!" << std::endl;
  return true;
}

template void FluidSolver<2>::save_checkpoint(const int);
template void FluidSolver<3>::save_checkpoint(const int);
template bool FluidSolver<2>::load_checkpoint();
template bool FluidSolver<3>::load_checkpoint();

} // namespace MPI
} // namespace Fluid
} // namespace ifem_host
