#include "parameters.hpp"
#include <algorithm>
#include <fstream>
#include <sstream>
#include <stdexcept>

namespace ifem_host {
namespace Parameters {

static std::string trim(const std::string &s) {
  const size_t a = s.find_first_not_of(" \t\r\n");
  if (a == std::string::npos) return "";
  const size_t b = s.find_last_not_of(" \t\r\n");
  return s.substr(a, b - a + 1);
}

template <class T>
static std::vector<T> parse_list(const std::string &s) {
  std::vector<T> out;
  std::stringstream ss(s);
  std::string tok;
  while (std::getline(ss, tok, ',')) {
    tok = trim(tok);
    if (tok.empty()) continue;
    std::stringstream ts(tok);
    T v;
    ts >> v;
    if (ts.fail()) throw std::invalid_argument("cannot parse list entry '" + tok + "'");
    out.push_back(v);
  }
  return out;
}

AllParameters AllParameters::from_string(const std::string &text) {
  std::map<std::string, std::map<std::string, std::string>> kv;
  std::stringstream in(text);
  std::string line, section;
  while (std::getline(in, line)) {
    const size_t hash = line.find('#');
    if (hash != std::string::npos) line = line.substr(0, hash);
    line = trim(line);
    if (line.empty()) continue;
    if (line.rfind("subsection", 0) == 0) section = trim(line.substr(10));
    else if (line == "end") section.clear();
    else if (line.rfind("set", 0) == 0) {
      const size_t eq = line.find('=');
      if (eq == std::string::npos) throw std::invalid_argument("malformed line: " + line);
      kv[section][trim(line.substr(3, eq - 3))] = trim(line.substr(eq + 1));
    } else
      throw std::invalid_argument("unrecognised line in parameter file: " + line);
  }
  auto has = [&](const std::string &s, const std::string &k) { return kv.count(s) && kv[s].count(k); };
  auto getd = [&](const std::string &s, const std::string &k, double def) { return has(s, k) ? std::stod(kv[s][k]) : def; };
  auto geti = [&](const std::string &s, const std::string &k, int def) { return has(s, k) ? std::stoi(kv[s][k]) : def; };
  auto gets = [&](const std::string &s, const std::string &k, const std::string &def) { return has(s, k) ? kv[s][k] : def; };
  AllParameters p;
  p.simulation_type = gets("Simulation", "Simulation type", "FSI");
  p.dimension = geti("Simulation", "Dimension", 2);
  p.global_refinements = parse_list<int>(gets("Simulation", "Global refinements", "0, 0"));
  if (p.global_refinements.size() != 2) throw std::invalid_argument("Incorrect dimension of global_refinements!");
  p.end_time = getd("Simulation", "End time", 1.0);
  p.time_step = getd("Simulation", "Time step size", 1.0);
  p.output_interval = getd("Simulation", "Output interval", 1.0);
  p.refinement_interval = getd("Simulation", "Refinement interval", 1.0);
  p.save_interval = getd("Simulation", "Save interval", 1.0);
  p.gravity = parse_list<double>(gets("Simulation", "Gravity", ""));
  if ((int)p.gravity.size() != p.dimension) throw std::invalid_argument("Inconsistent dimension of gravity!");
  p.fluid_pressure_degree = geti("Fluid finite element system", "Pressure degree", 1);
  p.fluid_velocity_degree = geti("Fluid finite element system", "Velocity degree", 2);
  p.viscosity = getd("Fluid material properties", "Dynamic viscosity", 1e-3);
  p.fluid_rho = getd("Fluid material properties", "Fluid density", 1.0);
  p.solid_rho = getd("Solid material properties", "Solid density", 1.0);
  p.grad_div = getd("Fluid solver control", "Grad-Div stabilization", 1.0);
  p.fluid_max_iterations = geti("Fluid solver control", "Max Newton iterations", 8);
  p.fluid_tolerance = getd("Fluid solver control", "Nonlinear system tolerance", 1e-10);
  const std::string D = "Fluid Dirichlet BCs";
  p.use_hard_coded_values = geti(D, "Use hard-coded boundary values", 0);
  p.n_fluid_dirichlet_bcs = geti(D, "Number of Dirichlet BCs", 0);
  const auto ids = parse_list<int>(gets(D, "Dirichlet boundary id", ""));
  const auto comps = parse_list<int>(gets(D, "Dirichlet boundary components", ""));
  const auto vals = parse_list<double>(gets(D, "Dirichlet boundary values", ""));
  if (p.n_fluid_dirichlet_bcs && ids.size() != p.n_fluid_dirichlet_bcs) throw std::invalid_argument("Inconsistent boundary ids!");
  if (p.n_fluid_dirichlet_bcs && comps.size() != p.n_fluid_dirichlet_bcs) throw std::invalid_argument("Inconsistent boundary components!");
  size_t n = 0;
  for (unsigned i = 0; i < p.n_fluid_dirichlet_bcs; ++i) {
    const int flag = comps[i];
    if (flag < 1 || flag > 7) throw std::invalid_argument("Dirichlet boundary components must be in 1..7");
    if (n >= vals.size()) throw std::invalid_argument("Inconsistent boundary values!");
    const int cnt = (flag == 1 || flag == 2 || flag == 4) ? 1 : ((flag == 3 || flag == 5 || flag == 6) ? 2 : 3);
    if (n + cnt > vals.size()) throw std::invalid_argument("Inconsistent boundary values!");
    std::vector<double> v(vals.begin() + n, vals.begin() + n + cnt);
    n += cnt;
    p.fluid_dirichlet_bcs[(unsigned)ids[i]] = {(unsigned)flag, v};
  }
  if (p.n_fluid_dirichlet_bcs && n != vals.size()) throw std::invalid_argument("Inconsistent boundary values!");
  const std::string N = "Fluid Neumann BCs";
  p.n_fluid_neumann_bcs = geti(N, "Number of Neumann BCs", 0);
  const auto nids = parse_list<int>(gets(N, "Neumann boundary id", ""));
  const auto nvals = parse_list<double>(gets(N, "Neumann boundary values", ""));
  if (p.n_fluid_neumann_bcs && nids.size() != p.n_fluid_neumann_bcs) throw std::invalid_argument("Inconsistent boundary ids!");
  if (p.n_fluid_neumann_bcs && nvals.size() != p.n_fluid_neumann_bcs) throw std::invalid_argument("Inconsistent boundary values!");
  for (unsigned i = 0; i < p.n_fluid_neumann_bcs; ++i) p.fluid_neumann_bcs[(unsigned)nids[i]] = nvals[i];
  if (p.fluid_velocity_degree < 1 || p.fluid_velocity_degree > 2 || p.fluid_pressure_degree != 1)
    throw std::invalid_argument("supported elements: Q2/Q1 and Q1/Q1");
  return p;
}

AllParameters::AllParameters(const std::string &infile) {
  std::ifstream f(infile);
  if (!f) throw std::invalid_argument("cannot open parameter file " + infile);
  std::stringstream ss;
  ss << f.rdbuf();
  *this = from_string(ss.str());
}

} // namespace Parameters
} // namespace ifem_host
