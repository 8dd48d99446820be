// parameters.hpp -- the Fluid/Simulation subset of Parameters::AllParameters (reference include/parameters.h,
// source/parameters.cpp:7-288) read from the same deal.II-style .prm files the reference tests ship.
#pragma once
#include <map>
#include <string>
#include <vector>

namespace ifem_host {
namespace Parameters {

struct AllParameters {
  // subsection Simulation (parameters.cpp:7-73)
  std::string simulation_type = "Fluid";
  int dimension = 2;
  std::vector<int> global_refinements{0, 0};
  double end_time = 1.0, time_step = 1.0, output_interval = 1.0, refinement_interval = 1.0, save_interval = 1.0;
  std::vector<double> gravity;
  // subsection Fluid finite element system (:76-100)
  unsigned fluid_pressure_degree = 1, fluid_velocity_degree = 2;
  // subsection Fluid material properties (:102-124)
  double viscosity = 1e-3, fluid_rho = 1.0;
  // subsection Solid material properties (:386-435): density of the artificial fluid in SCnsIM (mpi_scnsim.cpp:210-213)
  double solid_rho = 1.0;
  // subsection Fluid solver control (:126-156)
  double grad_div = 1.0;
  unsigned fluid_max_iterations = 8;
  double fluid_tolerance = 1e-10;
  // subsection Fluid Dirichlet BCs (:158-241): id -> (component flag, values)
  int use_hard_coded_values = 0;
  unsigned n_fluid_dirichlet_bcs = 0;
  std::map<unsigned, std::pair<unsigned, std::vector<double>>> fluid_dirichlet_bcs;
  // subsection Fluid Neumann BCs (:243-288): id -> pressure
  unsigned n_fluid_neumann_bcs = 0;
  std::map<unsigned, double> fluid_neumann_bcs;

  AllParameters() = default;
  explicit AllParameters(const std::string &infile);
  // parse from text (same grammar: "subsection X" / "set Key = value" / "end", '#' comments)
  static AllParameters from_string(const std::string &text);
};

} // namespace Parameters
} // namespace ifem_host
