"""MI355X-native ADMM quadratic-program solver: drop-in for the SolveQuadraticProgram.jl + LinearSystemSolvers.jl path
of RoyiAvital/QuadraticProgramSolver.  The compute lives in libqps_hip.so (hand-written HIP for gfx950, C ABI in
include/qps.h); this package is the host-side mirror of the reference interface plus the GenerateRandomQP harness."""
from .generator import (GenerateDenseBenchmarkQP, GenerateRandomQP, GenerateSparseBenchmarkQP, LoadQpModel, ProblemClass,
                        SaveQpModel, make_rng, sprandn)
from .solver import (AutoLinearSolverMode, ConvergenceFlag, HipCg, HipCgInit, HipItrSolCg, HipItrSolCgInit, HipLdl, HipLdlInit, HipChol, HipCholF32, HipCholF32Init, HipCholInit,
                     LinearSolverMode, QuadraticProgram, QuadraticProgramBatch, SolveQuadraticProgram, SolveQuadraticProgram_b,
                     SolveQuadraticProgramInplace)
from .proxqp import ProxQP, SolveQuadraticProgramProxQP
from ._lib import QpsError, QpsLibraryError

__all__ = ["GenerateRandomQP", "GenerateDenseBenchmarkQP", "GenerateSparseBenchmarkQP", "ProblemClass", "make_rng",
           "sprandn", "SaveQpModel", "LoadQpModel", "ConvergenceFlag", "LinearSolverMode", "QuadraticProgram", "QuadraticProgramBatch", "SolveQuadraticProgram",
           "SolveQuadraticProgramInplace", "SolveQuadraticProgram_b", "HipCholInit", "HipChol", "HipCgInit", "HipCg", "HipItrSolCgInit", "HipItrSolCg", "HipLdlInit", "HipLdl", "AutoLinearSolverMode",
           "HipCholF32Init", "HipCholF32", "ProxQP", "SolveQuadraticProgramProxQP", "QpsError", "QpsLibraryError"]
