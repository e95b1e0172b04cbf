# QuadraticProgramSolverHIP.jl -- thin `ccall` layer over libqps_hip.so (C ABI: include/qps.h).
#
# NOT EXECUTED in this pipeline: neither the build container nor the GPU box has Julia.  All logic lives below the C
# ABI; this file only marshals arguments, so that it can be reviewed by reading.  A maintainer of the reference drops it
# next to SolveQuadraticProgram.jl and passes the sentinel pair to the unmodified call:
#
#     include("QuadraticProgramSolverHIP.jl")
#     convFlag = SolveQuadraticProgram!(vX, mP, vQ, mA, vL, vU, HipCholInit, HipChol!; ρ = 0.1, adptΡ = true)
#
# `SolveQuadraticProgram!` below is a method specialised on the sentinel types, so the reference's generic method
# (SolveQuadraticProgram.jl:14) keeps serving every other plugin pair.

const LIBQPS = get(ENV, "QPS_HIP_LIB", joinpath(@__DIR__, "..", "quadraticprogramsolver_amd", "libqps_hip.so"))

# mirrors qps_params / qps_info of include/qps.h field by field
struct QpsParams
    numIterations::Int32; adptRho::Int32; numItrConv::Int32; numItrPolish::Int32; numItrMinres::Int32
    linsys::Int32; trsvBlock::Int32; reuseFactor::Int32
    epsAbs::Float64; epsRel::Float64; rho::Float64; sigma::Float64; alpha::Float64; delta::Float64
    fctrRho::Float64; epsMinres::Float64; epsPcg::Float64
    numItrPcg::Int32; loopVariant::Int32; polish::Int32; reserved0::Int32
end
mutable struct QpsInfo
    convFlag::Int32; iterations::Int32; numRefactor::Int32; cgIterations::Int32
    rhoFinal::Float64; rhoProposed::Float64; resPrim::Float64; resDual::Float64
    tSetup::Float64; tLoop::Float64; tRefactor::Float64
    polishFlag::Int32; polishIterations::Int32; tPolish::Float64
    trsvBlock::Int32; sweepVariant::Int32
    QpsInfo() = new(0, 0, 0, 0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, -1, 0, 0.0, 0, 0)
end

struct HipCholInitT end;  const HipCholInit = HipCholInitT()      # dense reduced-form Cholesky on the device
struct HipCholT end;      const HipChol! = HipCholT()
struct HipCgInitT end;    const HipCgInit = HipCgInitT()          # CSR matrix-free CG on the device
struct HipCgT end;        const HipCg! = HipCgT()
struct HipLdlInitT end;   const HipLdlInit = HipLdlInitT()        # sparse L D L' of the KKT matrix on the device: the counterpart of
struct HipLdlT end;       const HipLdl! = HipLdlT()               # LaLdlInit/LaLdl!, QDLdlInit/QDLdl!, FacLdlInit/FacLdl! (LinearSystemSolvers.jl:16-107)
const QPS_LINSYS_CHOLESKY = Int32(1); const QPS_LINSYS_CG = Int32(2); const QPS_LINSYS_KKT_LDL = Int32(3)

function _check(status::Int32, h::Ptr{Cvoid} = C_NULL)
    status == 0 && return
    msg = unsafe_string(ccall((:qps_last_error, LIBQPS), Cstring, (Ptr{Cvoid},), h))
    status in (1, 2, 3) ? throw(ArgumentError(msg)) : error(msg)
end

# SparseMatrixCSC{Float64,Int64} fields go through unchanged (colptr/rowval/nzval, index_base = 1)
function _create(mP::SparseMatrixCSC{Float64,Int64}, vQ, mA::SparseMatrixCSC{Float64,Int64}, vL, vU; densePath::Bool, device = 0)
    h = Ref{Ptr{Cvoid}}(C_NULL)
    n, m = size(mP, 1), size(mA, 1)
    GC.@preserve mP vQ mA vL vU begin
        _check(ccall((:qps_create_csc, LIBQPS), Int32,
            (Int64, Int64, Ptr{Int64}, Ptr{Int64}, Ptr{Float64}, Ptr{Int64}, Ptr{Int64}, Ptr{Float64},
             Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Int32, Int32, Int32, Int32, Ref{Ptr{Cvoid}}),
            n, m, mP.colptr, mP.rowval, mP.nzval, mA.colptr, mA.rowval, mA.nzval, vQ, vL, vU,
            Int32(1), Int32(densePath), Int32(0), Int32(device), h))
    end
    return h[]
end
function _create(mP::Matrix{Float64}, vQ, mA::Matrix{Float64}, vL, vU; densePath::Bool = true, device = 0)
    h = Ref{Ptr{Cvoid}}(C_NULL)
    n, m = size(mP, 1), size(mA, 1)
    GC.@preserve mP vQ mA vL vU begin
        _check(ccall((:qps_create_dense, LIBQPS), Int32,
            (Int64, Int64, Ptr{Float64}, Int64, Ptr{Float64}, Int64, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Int32, Int32, Ref{Ptr{Cvoid}}),
            n, m, mP, stride(mP, 2), mA, max(stride(mA, 2), 1), vQ, vL, vU, Int32(0), Int32(device), h))
    end
    return h[]
end

function _solve!(vX::Vector{Float64}, mP, vQ, mA, vL, vU, densePath::Bool, linsys::Int32 = densePath ? QPS_LINSYS_CHOLESKY : QPS_LINSYS_CG;
    numIterations = 5000, ϵAbs = 1e-6, ϵRel = 1e-6, ρ = 1, σ = 1e-6, α = 1.6, δ = 1e-6, adptΡ::Bool = false,
    fctrΡ = 5, numItrConv = 25, numItrPolish = 10, ϵMinres = 1e-6, numItrMinres = 500, info = nothing,
    polish::Bool = false)   # polish = true: the polishing step of SolveQuadraticProgram.m:289-325 (the Julia loop reserves its kwargs unused)
    h = _create(mP, Vector{Float64}(vQ), mA, Vector{Float64}(vL), Vector{Float64}(vU); densePath = densePath)
    try
        prm = QpsParams(numIterations, adptΡ, numItrConv, numItrPolish, numItrMinres, linsys, 0, 0,
                        ϵAbs, ϵRel, ρ, σ, α, δ, fctrΡ, ϵMinres, 1e-6, 1000, 0, polish, 0)
        inf = QpsInfo()
        GC.@preserve vX _check(ccall((:qps_solve, LIBQPS), Int32, (Ptr{Cvoid}, Ptr{Float64}, Ref{QpsParams}, Ref{QpsInfo}),
                                     h, vX, Ref(prm), inf), h)
        info === nothing || (info[] = inf)
        return ConvergenceFlag(inf.convFlag)     # the reference's own enum (SolveQuadraticProgram.jl:12)
    finally
        ccall((:qps_destroy, LIBQPS), Int32, (Ptr{Cvoid},), h)
    end
end

# Same positional order and keyword names as SolveQuadraticProgram.jl:14-17
SolveQuadraticProgram!(vX, mP, vQ, mA, vL, vU, ::HipCholInitT, ::HipCholT; kw...) = _solve!(vX, mP, vQ, mA, vL, vU, true; kw...)
SolveQuadraticProgram!(vX, mP, vQ, mA, vL, vU, ::HipCgInitT, ::HipCgT; kw...) = _solve!(vX, mP, vQ, mA, vL, vU, false; kw...)
# RunTests.jl:55-56 / RunBenchmarks.jl:54-55 select FacLdlInit / FacLdl!; the device counterpart takes the same SparseMatrixCSC inputs
SolveQuadraticProgram!(vX, mP::SparseMatrixCSC, vQ, mA::SparseMatrixCSC, vL, vU, ::HipLdlInitT, ::HipLdlT; kw...) =
    _solve!(vX, mP, vQ, mA, vL, vU, false, QPS_LINSYS_KKT_LDL; kw...)

# modeAuto of SolveQuadraticProgramRef! (SolveQuadraticProgram.jl:143-151), evaluated by the library so that every binding shares one rule
function AutoLinearSystemPair(mP, mA)
    sparseIn = (mP isa SparseMatrixCSC) && (mA isa SparseMatrixCSC)
    kind = ccall((:qps_linsys_auto, LIBQPS), Int32, (Int64, Int64, Int64, Int64, Int32), size(mP, 1), size(mA, 1), nnz(sparse(mP)), nnz(sparse(mA)), Int32(sparseIn))
    return kind == QPS_LINSYS_CG ? (HipCgInit, HipCg!) : (kind == QPS_LINSYS_KKT_LDL ? (HipLdlInit, HipLdl!) : (HipCholInit, HipChol!))
end

# Convenience form named in the project brief: SolveQuadraticProgram(P, q, A, l, u; ...) -> (x, flag)
function SolveQuadraticProgram(mP, vQ, mA, vL, vU; kw...)
    vX = zeros(size(mP, 1))
    flag = SolveQuadraticProgram!(vX, mP, vQ, mA, vL, vU, HipCholInit, HipChol!; kw...)
    return vX, flag
end

# The plugin pair called literally by the UNMODIFIED reference loop (SolveQuadraticProgram.jl:36,54): host vectors in,
# device linear solve, host vectors out.  tuSolver holds the handle; a finaliser destroys it.
mutable struct HipLinSys
    h::Ptr{Cvoid}
    function HipLinSys(h)
        s = new(h)
        finalizer(x -> ccall((:qps_destroy, LIBQPS), Int32, (Ptr{Cvoid},), x.h), s)
    end
end
function (::HipCholInitT)(vX, mP, vQ, mA, ρ, ρ¹, σ, numElements, numConstraints)
    h = _create(mP, Vector{Float64}(vQ), mA, zeros(numConstraints), zeros(numConstraints); densePath = true)
    _check(ccall((:qps_linsys_init, LIBQPS), Int32, (Ptr{Cvoid}, Float64, Float64, Int32, Int32), h, ρ, σ, 1, 0), h)
    return zeros(numElements), zeros(numConstraints), Any[HipLinSys(h)]
end
# the direct KKT plugins' literal signature (LinearSystemSolvers.jl:16,28): Init factorises [mP + σI  mA'; mA  -ρ¹I] on the device (ordering +
# symbolic once), Sol! re-factorises numerically on changedΡ and solves
function (::HipLdlInitT)(vX, mP::SparseMatrixCSC, vQ, mA::SparseMatrixCSC, ρ, ρ¹, σ, numElements, numConstraints)
    h = _create(mP, Vector{Float64}(vQ), mA, zeros(numConstraints), zeros(numConstraints); densePath = false)
    _check(ccall((:qps_linsys_init, LIBQPS), Int32, (Ptr{Cvoid}, Float64, Float64, Int32, Int32), h, ρ, σ, QPS_LINSYS_KKT_LDL, 0), h)
    return zeros(numElements), zeros(numConstraints), Any[HipLinSys(h)]
end
(::HipLdlT)(tuSolver, vXX, vZZ, vX, mP, vQ, mA, vZ, vY, ρ, ρ¹, σ, numElements, numConstraints, changedΡ) =
    HipChol!(tuSolver, vXX, vZZ, vX, mP, vQ, mA, vZ, vY, ρ, ρ¹, σ, numElements, numConstraints, changedΡ)   # same qps_linsys_solve call

function (::HipCholT)(tuSolver, vXX, vZZ, vX, mP, vQ, mA, vZ, vY, ρ, ρ¹, σ, numElements, numConstraints, changedΡ)
    h = tuSolver[1].h
    GC.@preserve vXX vZZ vX vZ vY _check(ccall((:qps_linsys_solve, LIBQPS), Int32,
        (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Float64, Float64, Int32, Ptr{Float64}, Ptr{Float64}),
        h, vX, vZ, vY, ρ, σ, Int32(changedΡ), vXX, vZZ), h)
    return
end


# ---------------------------------------------------------------------------------------------------------------------
# The second solver form (ProxQP.jl):  min 1/2 x'Px + q'x  s.t.  A x = b,  C x <= d   -- device version (dense inputs).
# Mirrors `ProxQP(mP, vQ, mA, vB, mC, vD)` (ProxQP.jl:73-93) and `SolveQuadraticProgram!(sQpProb; ...)` (:118-173).
# ---------------------------------------------------------------------------------------------------------------------
struct QpsProxQpParams
    numIterations::Int32; numItrConv::Int32; adptRho::Int32; loopVariant::Int32
    epsAbs::Float64; epsRel::Float64; rho::Float64; sigma::Float64; tau::Float64
end
mutable struct QpsProxQpReport
    converged::Int32; iterations::Int32; rho::Float64; sigma::Float64; resPrim::Float64; resDual::Float64
    QpsProxQpReport() = new(0, 0, 0.0, 0.0, 0.0, 0.0)
end
mutable struct ProxQPHip{T <: AbstractFloat}
    h  :: Ptr{Cvoid}
    vX :: Vector{T}; vY :: Vector{T}; vZ :: Vector{T}; vS :: Vector{T}     # same field names as ProxQP.jl:9-18
    dataDim :: Int; numEq :: Int; numInEq :: Int
end
function ProxQPHip(mP :: Matrix{Float64}, vQ :: Vector{Float64}, mA :: Matrix{Float64}, vB :: Vector{Float64}, mC :: Matrix{Float64}, vD :: Vector{Float64})
    n, me, mi = size(mP, 1), size(mA, 1), size(mC, 1)
    h = Ref{Ptr{Cvoid}}(C_NULL)
    GC.@preserve mP vQ mA vB mC vD _check(ccall((:qps_proxqp_create_dense, LIBQPS), Int32,
        (Int64, Int64, Int64, Ptr{Float64}, Int64, Ptr{Float64}, Ptr{Float64}, Int64, Ptr{Float64}, Ptr{Float64}, Int64, Ptr{Float64}, Int32, Int32, Ref{Ptr{Cvoid}}),
        n, me, mi, mP, n, vQ, mA, max(me, 1), vB, mC, max(mi, 1), vD, Int32(0), Int32(0), h))
    _check(ccall((:qps_proxqp_init_kkt, LIBQPS), Int32, (Ptr{Cvoid},), h[]), h[])          # ProxQP.jl:80-89 on the device
    s = ProxQPHip{Float64}(h[], zeros(n), zeros(max(me, 1)), zeros(max(mi, 1)), zeros(max(mi, 1)), n, me, mi)
    finalizer(x -> ccall((:qps_destroy, LIBQPS), Int32, (Ptr{Cvoid},), x.h), s)
    return s
end
# SparseProxQP (ProxQP.jl:71, :95-115): the colptr / rowval / nzval fields as they are (1-based Int64); the matrices stay sparse on the device
function ProxQPHip(mP :: SparseMatrixCSC{Float64, Int64}, vQ :: Vector{Float64}, mA :: SparseMatrixCSC{Float64, Int64}, vB :: Vector{Float64},
                   mC :: SparseMatrixCSC{Float64, Int64}, vD :: Vector{Float64})
    n, me, mi = size(mP, 1), size(mA, 1), size(mC, 1)
    h = Ref{Ptr{Cvoid}}(C_NULL)
    vBp = isempty(vB) ? zeros(1) : vB; vDp = isempty(vD) ? zeros(1) : vD
    GC.@preserve mP vQ mA vBp mC vDp _check(ccall((:qps_proxqp_create_csc, LIBQPS), Int32,
        (Int64, Int64, Int64, Ptr{Int64}, Ptr{Int64}, Ptr{Float64}, Ptr{Float64}, Ptr{Int64}, Ptr{Int64}, Ptr{Float64}, Ptr{Float64},
         Ptr{Int64}, Ptr{Int64}, Ptr{Float64}, Ptr{Float64}, Int32, Int32, Int32, Ref{Ptr{Cvoid}}),
        n, me, mi, mP.colptr, mP.rowval, mP.nzval, vQ, mA.colptr, mA.rowval, mA.nzval, vBp, mC.colptr, mC.rowval, mC.nzval, vDp,
        Int32(1), Int32(0), Int32(0), h))
    _check(ccall((:qps_proxqp_init_kkt, LIBQPS), Int32, (Ptr{Cvoid},), h[]), h[])          # ProxQP.jl:102-111 on the device
    s = ProxQPHip{Float64}(h[], zeros(n), zeros(max(me, 1)), zeros(max(mi, 1)), zeros(max(mi, 1)), n, me, mi)
    finalizer(x -> ccall((:qps_destroy, LIBQPS), Int32, (Ptr{Cvoid},), x.h), s)
    return s
end
function SolveQuadraticProgram!(sQpProb :: ProxQPHip{Float64}; numIterations = 2000, ϵAbs = 1e-7, ϵRel = 1e-6, numItrConv = 50,
                                ρ = 1e2, σ = 1e-2, adptΡ :: Bool = true, τ = 10.0)
    prm = QpsProxQpParams(numIterations, numItrConv, adptΡ, 0, ϵAbs, ϵRel, ρ, σ, τ)
    rep = QpsProxQpReport()
    _check(ccall((:qps_proxqp_solve, LIBQPS), Int32, (Ptr{Cvoid}, Ref{QpsProxQpParams}, Ref{QpsProxQpReport}), sQpProb.h, Ref(prm), rep), sQpProb.h)
    GC.@preserve sQpProb _check(ccall((:qps_proxqp_get_state, LIBQPS), Int32, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}),
                                      sQpProb.h, sQpProb.vX, sQpProb.vY, sQpProb.vZ, sQpProb.vS), sQpProb.h)
    return Dict{String, Real}("Converged" => rep.converged != 0, "Iterations" => rep.iterations, "ρ" => rep.rho, "σ" => rep.sigma,
                              "PrimalResidual" => rep.resPrim, "DualResidual" => rep.resDual)     # ProxQP.jl:127
end


# ---------------------------------------------------------------------------------------------------------------------
# Polishing step of the MATLAB implementation (SolveQuadraticProgram.m:289-325) on the device: `polish = true` in
# `SolveQuadraticProgram!` chains it after the loop; `PolishQuadraticProgram!` runs it alone on a given (vX, vY).
# ---------------------------------------------------------------------------------------------------------------------
mutable struct QpsPolishReport
    flag::Int32; refinements::Int32; minresIterations::Int32; numActiveLower::Int32; numActiveUpper::Int32; reserved0::Int32
    relres::Float64; seconds::Float64
    QpsPolishReport() = new(-1, 0, 0, 0, 0, 0, NaN, 0.0)
end
function PolishQuadraticProgram!(vX::Vector{Float64}, vY::Vector{Float64}, mP, vQ, mA, vL, vU;
    numItrPolish = 10, δ = 1e-6, ϵMinres = 1e-6, numItrMinres = 500, densePath::Bool = true)
    h = _create(mP, Vector{Float64}(vQ), mA, Vector{Float64}(vL), Vector{Float64}(vU); densePath = densePath)
    try
        prm = QpsParams(5000, false, 25, numItrPolish, numItrMinres, densePath ? 1 : 2, 0, 0,
                        1e-6, 1e-6, 1.0, 1e-6, 1.6, δ, 5.0, ϵMinres, 1e-6, 1000, 0, 1, 0)
        rep = QpsPolishReport()
        GC.@preserve vX vY _check(ccall((:qps_polish, LIBQPS), Int32, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ref{QpsParams}, Ref{QpsPolishReport}),
                                        h, vX, vY, Ref(prm), rep), h)
        return rep          # rep.flag == 0: vX was replaced by the polished primal (minresFlag semantics of :311-325)
    finally
        ccall((:qps_destroy, LIBQPS), Int32, (Ptr{Cvoid},), h)
    end
end

# ---------------------------------------------------------------------------------------------------------------------
# A batch of independent dense QPs of one shape (the per-problem loop of RunBenchmarks.jl:88-104 advanced in lock step).
# mX is n x count (one column per problem: warm starts in, solutions out).
# ---------------------------------------------------------------------------------------------------------------------
function SolveQuadraticProgramBatch!(mX::Matrix{Float64}, vmP::Vector{Matrix{Float64}}, mQ::Matrix{Float64}, vmA::Vector{Matrix{Float64}},
    mL::Matrix{Float64}, mU::Matrix{Float64}; numIterations = 5000, ϵAbs = 1e-6, ϵRel = 1e-6, ρ = 1, σ = 1e-6, α = 1.6,
    adptΡ::Bool = false, fctrΡ = 5, numItrConv = 25, device = 0)
    count = length(vmP); n = size(vmP[1], 1); m = size(vmA[1], 1)
    P = reduce(hcat, vec.(vmP)); A = m > 0 ? reduce(hcat, vec.(vmA)) : zeros(1, count)     # column-major matrices back to back
    h = Ref{Ptr{Cvoid}}(C_NULL)
    GC.@preserve P A mQ mL mU _check(ccall((:qps_create_dense_batch, LIBQPS), Int32,
        (Int64, Int64, Int64, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Int32, Int32, Ref{Ptr{Cvoid}}),
        count, n, m, P, A, mQ, mL, mU, Int32(0), Int32(device), h))
    try
        prm = QpsParams(numIterations, adptΡ, numItrConv, 10, 500, 0, 0, 0, ϵAbs, ϵRel, ρ, σ, α, 1e-6, fctrΡ, 1e-6, 1e-6, 1000, 0, 0, 0)
        buf = Vector{UInt8}(undef, count * sizeof(QpsInfo))
        GC.@preserve mX buf _check(ccall((:qps_solve_batch, LIBQPS), Int32, (Ptr{Cvoid}, Ptr{Float64}, Ref{QpsParams}, Ptr{UInt8}),
                                         h[], mX, Ref(prm), buf), h[])
        flags = [ConvergenceFlag(unsafe_load(Ptr{Int32}(pointer(buf) + (b - 1) * sizeof(QpsInfo)))) for b in 1:count]
        return flags
    finally
        ccall((:qps_destroy, LIBQPS), Int32, (Ptr{Cvoid},), h[])
    end
end
