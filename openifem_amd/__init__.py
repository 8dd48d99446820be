"""openifem_amd -- MI355X-native implementation of OpenIFEM's implicit Navier-Stokes fluid step.

The product is lib/libifem_hip.so (C ABI in include/ifem_hip.h) plus the C++ host mirror of
Fluid::MPI::FluidSolver / InsIM in csrc/host/.  This package only binds them for tests and bench.py.
"""
from . import capi  # noqa: F401
