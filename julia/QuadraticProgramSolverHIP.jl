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
    trsvBlock::Int32; sweepVariant::Int32; sweepGaveUp::Int32; cgExplicit::Int32
    QpsInfo() = new(0, 0, 0, 0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, -1, 0, 0.0, 0, 0, 0, 0)
end

struct HipCholInitT end;  const HipCholInit = HipCholInitT()      # dense reduced-form Cholesky on the device
struct HipCholT end;      const HipChol! = HipCholT()
struct HipCgInitT end;    const HipCgInit = HipCgInitT()          # CSR matrix-free CG on the device
struct HipCgT end;        const HipCg! = HipCgT()
struct HipLdlInitT end;   const HipLdlInit = HipLdlInitT()        # sparse L D L' of the KKT matrix on the device: the counterpart of
struct HipLdlT end;       const HipLdl! = HipLdlT()               # LaLdlInit/LaLdl!, QDLdlInit/QDLdl!, FacLdlInit/FacLdl! (LinearSystemSolvers.jl:16-107)
struct HipItrSolCgInitT end; const HipItrSolCgInit = HipItrSolCgInitT()   # cg! on the EXPLICIT reduced matrix mPI + ρ mAA on the device: the counterpart of
struct HipItrSolCgT end;     const HipItrSolCg! = HipItrSolCgT()          # ItrSolCgInit / ItrSolCg! (LinearSystemSolvers.jl:110-142)
const QPS_LINSYS_CHOLESKY = Int32(1); const QPS_LINSYS_CG = Int32(2); const QPS_LINSYS_KKT_LDL = Int32(3); const QPS_LINSYS_CG_EXPLICIT = Int32(4)
const QPS_OP_P = Int32(0); const QPS_OP_A = Int32(1); const QPS_OP_AT = Int32(2); const QPS_OP_PA = Int32(3); const QPS_OP_REDUCED = Int32(4)   # qps_operator_kind
# arithmetic type of the device-resident loop (qps_dtype): the boundary always carries Float64 arrays, `dtype = Float32` runs the loop in fp32
_dtype(::Type{Float64}) = Int32(0)
_dtype(::Type{Float32}) = Int32(1)

function _check(status::Int32, h::Ptr{Cvoid} = C_NULL)
    status == 0 && return
    msg = unsafe_string(ccall((:qps_last_error, LIBQPS), Cstring, (Ptr{Cvoid},), h))
    status in (1, 2, 3) ? throw(ArgumentError(msg)) : error(msg)
end

# SparseMatrixCSC{Float64,Int64} fields go through unchanged (colptr/rowval/nzval, index_base = 1)
function _create(mP::SparseMatrixCSC{Float64,Int64}, vQ, mA::SparseMatrixCSC{Float64,Int64}, vL, vU; densePath::Bool, device = 0, dtype::Type = Float64)
    h = Ref{Ptr{Cvoid}}(C_NULL)
    n, m = size(mP, 1), size(mA, 1)
    GC.@preserve mP vQ mA vL vU begin
        _check(ccall((:qps_create_csc, LIBQPS), Int32,
            (Int64, Int64, Ptr{Int64}, Ptr{Int64}, Ptr{Float64}, Ptr{Int64}, Ptr{Int64}, Ptr{Float64},
             Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Int32, Int32, Int32, Int32, Ref{Ptr{Cvoid}}),
            n, m, mP.colptr, mP.rowval, mP.nzval, mA.colptr, mA.rowval, mA.nzval, vQ, vL, vU,
            Int32(1), Int32(densePath), _dtype(dtype), Int32(device), h))
    end
    return h[]
end
function _create(mP::Matrix{Float64}, vQ, mA::Matrix{Float64}, vL, vU; densePath::Bool = true, device = 0, dtype::Type = Float64)
    h = Ref{Ptr{Cvoid}}(C_NULL)
    n, m = size(mP, 1), size(mA, 1)
    GC.@preserve mP vQ mA vL vU begin
        _check(ccall((:qps_create_dense, LIBQPS), Int32,
            (Int64, Int64, Ptr{Float64}, Int64, Ptr{Float64}, Int64, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Int32, Int32, Ref{Ptr{Cvoid}}),
            n, m, mP, stride(mP, 2), mA, max(stride(mA, 2), 1), vQ, vL, vU, _dtype(dtype), Int32(device), h))
    end
    return h[]
end

function _solve!(vX::Vector{Float64}, mP, vQ, mA, vL, vU, densePath::Bool, linsys::Int32 = densePath ? QPS_LINSYS_CHOLESKY : QPS_LINSYS_CG;
    numIterations = 5000, ϵAbs = 1e-6, ϵRel = 1e-6, ρ = 1, σ = 1e-6, α = 1.6, δ = 1e-6, adptΡ::Bool = false,
    fctrΡ = 5, numItrConv = 25, numItrPolish = 10, ϵMinres = 1e-6, numItrMinres = 500, info = nothing,
    polish::Bool = false,   # polish = true: the polishing step of SolveQuadraticProgram.m:289-325 (the Julia loop reserves its kwargs unused)
    ϵPcg = 1e-6, numItrPcg = 1000,            # the CG plugins' own kwargs (LinearSystemSolvers.jl:125, :164, :207), used by (HipCgInit, HipCg!)
    dtype::Type = Float64, trsvBlock = 0, loopVariant = 0, device = 0)   # additive (qps_params / qps_create_*): fp32 loop, sweep block, loop variant, GPU
    if !densePath && !(mP isa SparseMatrixCSC && mA isa SparseMatrixCSC)     # the CG / L D L' plugins keep CSR storage: dense arrays are converted once
        mP = SparseMatrixCSC{Float64, Int64}(sparse(mP)); mA = SparseMatrixCSC{Float64, Int64}(sparse(mA))
    end
    h = _create(mP, Vector{Float64}(vQ), mA, Vector{Float64}(vL), Vector{Float64}(vU); densePath = densePath, device = device, dtype = dtype)
    try
        prm = QpsParams(numIterations, adptΡ, numItrConv, numItrPolish, numItrMinres, linsys, trsvBlock, 0,
                        ϵAbs, ϵRel, ρ, σ, α, δ, fctrΡ, ϵMinres, ϵPcg, numItrPcg, loopVariant, polish, 0)
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
SolveQuadraticProgram!(vX, mP, vQ, mA, vL, vU, ::HipItrSolCgInitT, ::HipItrSolCgT; kw...) = _solve!(vX, mP, vQ, mA, vL, vU, false, QPS_LINSYS_CG_EXPLICIT; kw...)
# RunTests.jl:55-56 / RunBenchmarks.jl:54-55 select FacLdlInit / FacLdl!; the device counterpart takes the same SparseMatrixCSC inputs
SolveQuadraticProgram!(vX, mP::SparseMatrixCSC, vQ, mA::SparseMatrixCSC, vL, vU, ::HipLdlInitT, ::HipLdlT; kw...) =
    _solve!(vX, mP, vQ, mA, vL, vU, false, QPS_LINSYS_KKT_LDL; kw...)

# modeAuto of SolveQuadraticProgramRef! (SolveQuadraticProgram.jl:143-151), evaluated by the library so that every binding shares one rule
function AutoLinearSystemPair(mP, mA)
    sparseIn = (mP isa SparseMatrixCSC) && (mA isa SparseMatrixCSC)
    kind = ccall((:qps_linsys_auto, LIBQPS), Int32, (Int64, Int64, Int64, Int64, Int32), size(mP, 1), size(mA, 1), nnz(sparse(mP)), nnz(sparse(mA)), Int32(sparseIn))
    return kind == QPS_LINSYS_CG ? (HipCgInit, HipCg!) : (kind == QPS_LINSYS_KKT_LDL ? (HipLdlInit, HipLdl!) : (HipCholInit, HipChol!))
end

# Convenience form named in the project brief: SolveQuadraticProgram(P, q, A, l, u; ...) -> (x, flag).  `linSolverMode` is the reference's own enum
# (SolveQuadraticProgram.jl:11, spelled as there: modeAuto, modeItertaive, modeDirect) and is resolved as SolveQuadraticProgramRef! resolves it
# (:135-151): modeItertaive -> matrix-free CG; modeDirect -> a factorisation (sparse L D L' of the KKT matrix for SparseMatrixCSC inputs, the dense
# reduced Cholesky otherwise); modeAuto -> the size / density rule, evaluated by the library (qps_linsys_auto).  The rule was tuned for a CPU:
# it sends every problem with more than 5000 rows to CG -- pass modeDirect to keep such a problem on the factorisation path.
function SolveQuadraticProgram(mP, vQ, mA, vL, vU; linSolverMode::LinearSolverMode = modeAuto, kw...)
    sparseIn = (mP isa SparseMatrixCSC) && (mA isa SparseMatrixCSC)
    pair = if linSolverMode == modeItertaive
        (HipCgInit, HipCg!)
    elseif linSolverMode == modeDirect
        sparseIn ? (HipLdlInit, HipLdl!) : (HipCholInit, HipChol!)
    else
        AutoLinearSystemPair(mP, mA)
    end
    vX = zeros(size(mP, 1))
    flag = SolveQuadraticProgram!(vX, mP, vQ, mA, vL, vU, pair...; kw...)
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
# the matrix-free CG plugins' literal signature: LinOpCgInit / LinOpCg! (LinearSystemSolvers.jl:145-186; same shape as ItrSolCg* :110-142 and
# LinMapsCg* :188-229).  Init keeps the CSC inputs as CSR on the device (the operator of :152-157 needs no factorisation); Sol! takes the
# reference's kwargs `ϵPcg, numItrPcg` (:164), runs the device-resident CG warm-started from the previous x~ (:179) and returns z~ = A x~ (:181).
function (::HipCgInitT)(vX, mP::SparseMatrixCSC, vQ, mA::SparseMatrixCSC, ρ, ρ¹, σ, numElements, numConstraints)
    h = _create(mP, Vector{Float64}(vQ), mA, zeros(numConstraints), zeros(numConstraints); densePath = false)
    _check(ccall((:qps_linsys_init, LIBQPS), Int32, (Ptr{Cvoid}, Float64, Float64, Int32, Int32), h, ρ, σ, QPS_LINSYS_CG, 0), h)
    return zeros(numElements), zeros(numConstraints), Any[HipLinSys(h)]
end
function (::HipCgT)(tuSolver, vXX, vZZ, vX, mP, vQ, mA, vZ, vY, ρ, ρ¹, σ, numElements, numConstraints, changedΡ; ϵPcg = 1e-6, numItrPcg = 1000)
    h = tuSolver[1].h
    _check(ccall((:qps_linsys_set_cg, LIBQPS), Int32, (Ptr{Cvoid}, Float64, Int32), h, ϵPcg, numItrPcg), h)
    HipChol!(tuSolver, vXX, vZZ, vX, mP, vQ, mA, vZ, vY, ρ, ρ¹, σ, numElements, numConstraints, changedΡ)   # same qps_linsys_solve call
    return
end

# ItrSolCgInit / ItrSolCg! literally (LinearSystemSolvers.jl:110-142): Init forms mAA = mA'mA, mPI = mP + σI and mL = mPI + ρ mAA on the device handle (:112-114),
# Sol! rebuilds mL from the cached parts on changedΡ (:127-129) and runs cg! with ONE product per iteration (:137); same kwargs as the other CG plugins (:125)
function (::HipItrSolCgInitT)(vX, mP::SparseMatrixCSC, vQ, mA::SparseMatrixCSC, ρ, ρ¹, σ, numElements, numConstraints)
    h = _create(mP, Vector{Float64}(vQ), mA, zeros(numConstraints), zeros(numConstraints); densePath = false)
    _check(ccall((:qps_linsys_init, LIBQPS), Int32, (Ptr{Cvoid}, Float64, Float64, Int32, Int32), h, ρ, σ, QPS_LINSYS_CG_EXPLICIT, 0), h)
    return zeros(numElements), zeros(numConstraints), Any[HipLinSys(h)]
end
(::HipItrSolCgT)(tuSolver, vXX, vZZ, vX, mP, vQ, mA, vZ, vY, ρ, ρ¹, σ, numElements, numConstraints, changedΡ; ϵPcg = 1e-6, numItrPcg = 1000) =
    HipCg!(tuSolver, vXX, vZZ, vX, mP, vQ, mA, vZ, vY, ρ, ρ¹, σ, numElements, numConstraints, changedΡ; ϵPcg = ϵPcg, numItrPcg = numItrPcg)   # same two calls

# One application of the operator's matrices on the device (qps_operator_apply): kind = QPS_OP_P / _A / _AT / _PA / _REDUCED -- mP v, mA v, mA' v, [mP; mA] v and
# (mP + ρ mA'mA + σI) v, the operator of LinOpCgInit (LinearSystemSolvers.jl:152-157) -- through the kernels the handle's solves use for those products
function ApplyOperator(tuSolver, kind::Int32, vV::Vector{Float64}; ρ = 1.0, σ = 0.0, numElements::Int, numConstraints::Int)
    h = tuSolver[1].h
    vOut = zeros(kind == QPS_OP_A ? numConstraints : (kind == QPS_OP_PA ? numElements + numConstraints : numElements))
    GC.@preserve vV vOut _check(ccall((:qps_operator_apply, LIBQPS), Int32, (Ptr{Cvoid}, Int32, Ptr{Float64}, Ptr{Float64}, Float64, Float64), h, kind, vV, vOut, ρ, σ), h)
    return vOut
end

# The per-problem loop of RunBenchmarks.jl:88-104 over the GPUs of this process (qps_solve_batch_multi): tProblems = vector of (mP, vQ, mA, vL, vU) dense tuples of one
# shape, vDevices = the device of every worker (one host thread each inside the library), chunk = the largest range a worker takes from the shared counter
# (0: one contiguous slab per worker, for fixed-K runs).  Returns (mX [n x count], vector of ConvergenceFlag, vector of QpsInfo, worker of every problem).
function SolveQuadraticProgramBatch(tProblems::Vector, vDevices::Vector{<:Integer}; chunk::Integer = 0, dtype::Type = Float64, numIterations = 5000, ϵAbs = 1e-6, ϵRel = 1e-6,
                                    ρ = 1, σ = 1e-6, α = 1.6, adptΡ::Bool = false, fctrΡ = 5, numItrConv = 25)
    count = length(tProblems); n = size(tProblems[1][1], 1); m = size(tProblems[1][3], 1)
    mPs = reduce(hcat, [vec(Matrix{Float64}(t[1])) for t in tProblems]); mAs = reduce(hcat, [vec(Matrix{Float64}(t[3])) for t in tProblems])   # column b = problem b, column-major
    mQ = reduce(hcat, [Vector{Float64}(t[2]) for t in tProblems]); mL = reduce(hcat, [Vector{Float64}(t[4]) for t in tProblems]); mU = reduce(hcat, [Vector{Float64}(t[5]) for t in tProblems])
    mX = zeros(n, count); vInfo = [QpsInfo() for _ in 1:count]; vWorker = zeros(Int32, count); vSeconds = zeros(length(vDevices))
    prm = QpsParams(numIterations, adptΡ, numItrConv, 10, 500, 0, 0, 0, ϵAbs, ϵRel, ρ, σ, α, 1e-6, fctrΡ, 1e-6, 1e-6, 1000, 0, 0, 0)
    vRaw = Vector{UInt8}(undef, count * sizeof(QpsInfo))                      # qps_info records land here (QpsInfo is mutable: not stored inline in a Vector)
    vDev = Vector{Int32}(vDevices)
    GC.@preserve mPs mAs mQ mL mU mX vRaw vDev vWorker vSeconds _check(ccall((:qps_solve_batch_multi, LIBQPS), Int32,
        (Int64, Int64, Int64, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Int32, Ptr{Int32}, Int32, Int32, Ptr{Float64}, Ref{QpsParams}, Ptr{UInt8}, Ptr{Int32}, Ptr{Float64}),
        count, n, m, mPs, mAs, mQ, mL, mU, _dtype(dtype), vDev, Int32(length(vDev)), Int32(chunk), mX, Ref(prm), vRaw, vWorker, vSeconds))
    for b in 1:count
        unsafe_copyto!(Ptr{UInt8}(pointer_from_objref(vInfo[b])), pointer(vRaw, (b - 1) * sizeof(QpsInfo) + 1), sizeof(QpsInfo))
    end
    return mX, [ConvergenceFlag(i.convFlag) for i in vInfo], vInfo, vWorker
end

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
# `ProxQP{T <: AbstractFloat}` (ProxQP.jl:8): T is the arithmetic type of the device-resident loop (Float64 or Float32); the C ABI carries Float64
# arrays either way, the state vectors come back as Vector{T}.
_f64(v::AbstractVector) = Vector{Float64}(v)
_f64(m::Matrix) = Matrix{Float64}(m)
_f64(m::SparseMatrixCSC) = SparseMatrixCSC{Float64, Int64}(m)
function ProxQPHip(mP :: Matrix{T}, vQ :: Vector{T}, mA :: Matrix{T}, vB :: Vector{T}, mC :: Matrix{T}, vD :: Vector{T}) where {T <: AbstractFloat}
    n, me, mi = size(mP, 1), size(mA, 1), size(mC, 1)
    h = Ref{Ptr{Cvoid}}(C_NULL)
    P, q, A, b, C, d = _f64(mP), _f64(vQ), _f64(mA), _f64(vB), _f64(mC), _f64(vD)
    GC.@preserve P q A b C d _check(ccall((:qps_proxqp_create_dense, LIBQPS), Int32,
        (Int64, Int64, Int64, Ptr{Float64}, Int64, Ptr{Float64}, Ptr{Float64}, Int64, Ptr{Float64}, Ptr{Float64}, Int64, Ptr{Float64}, Int32, Int32, Ref{Ptr{Cvoid}}),
        n, me, mi, P, n, q, A, max(me, 1), b, C, max(mi, 1), d, _dtype(T), Int32(0), h))
    _check(ccall((:qps_proxqp_init_kkt, LIBQPS), Int32, (Ptr{Cvoid},), h[]), h[])          # ProxQP.jl:80-89 on the device
    s = ProxQPHip{T}(h[], zeros(T, n), zeros(T, max(me, 1)), zeros(T, max(mi, 1)), zeros(T, max(mi, 1)), n, me, mi)
    finalizer(x -> ccall((:qps_destroy, LIBQPS), Int32, (Ptr{Cvoid},), x.h), s)
    _pull_state!(s)                                                                        # vX, vY, vS of the constructor (:80-89)
    return s
end
# SparseProxQP (ProxQP.jl:71, :95-115): the colptr / rowval / nzval fields as they are (1-based Int64); the matrices stay sparse on the device
function ProxQPHip(mP :: SparseMatrixCSC{T, Int64}, vQ :: Vector{T}, mA :: SparseMatrixCSC{T, Int64}, vB :: Vector{T},
                   mC :: SparseMatrixCSC{T, Int64}, vD :: Vector{T}) where {T <: AbstractFloat}
    n, me, mi = size(mP, 1), size(mA, 1), size(mC, 1)
    h = Ref{Ptr{Cvoid}}(C_NULL)
    P, q, A, C = _f64(mP), _f64(vQ), _f64(mA), _f64(mC)
    vBp = isempty(vB) ? zeros(1) : _f64(vB); vDp = isempty(vD) ? zeros(1) : _f64(vD)
    GC.@preserve P q A vBp C vDp _check(ccall((:qps_proxqp_create_csc, LIBQPS), Int32,
        (Int64, Int64, Int64, Ptr{Int64}, Ptr{Int64}, Ptr{Float64}, Ptr{Float64}, Ptr{Int64}, Ptr{Int64}, Ptr{Float64}, Ptr{Float64},
         Ptr{Int64}, Ptr{Int64}, Ptr{Float64}, Ptr{Float64}, Int32, Int32, Int32, Ref{Ptr{Cvoid}}),
        n, me, mi, P.colptr, P.rowval, P.nzval, q, A.colptr, A.rowval, A.nzval, vBp, C.colptr, C.rowval, C.nzval, vDp,
        Int32(1), _dtype(T), Int32(0), h))
    _check(ccall((:qps_proxqp_init_kkt, LIBQPS), Int32, (Ptr{Cvoid},), h[]), h[])          # ProxQP.jl:102-111 on the device
    s = ProxQPHip{T}(h[], zeros(T, n), zeros(T, max(me, 1)), zeros(T, max(mi, 1)), zeros(T, max(mi, 1)), n, me, mi)
    finalizer(x -> ccall((:qps_destroy, LIBQPS), Int32, (Ptr{Cvoid},), x.h), s)
    _pull_state!(s)
    return s
end
# device state -> sQpProb.vX / vY / vZ / vS (Float64 at the boundary, stored as T)
function _pull_state!(sQpProb :: ProxQPHip{T}) where {T <: AbstractFloat}
    x, y, z, sl = zeros(length(sQpProb.vX)), zeros(length(sQpProb.vY)), zeros(length(sQpProb.vZ)), zeros(length(sQpProb.vS))
    GC.@preserve x y z sl _check(ccall((:qps_proxqp_get_state, LIBQPS), Int32, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}),
                                       sQpProb.h, x, y, z, sl), sQpProb.h)
    sQpProb.vX .= x; sQpProb.vY .= y; sQpProb.vZ .= z; sQpProb.vS .= sl
    return sQpProb
end
# same keyword names, defaults and `T`-typed scalars as ProxQP.jl:118
function SolveQuadraticProgram!(sQpProb :: ProxQPHip{T}; numIterations :: Integer = 2000, ϵAbs = T(1e-7), ϵRel = T(1e-6), numItrConv :: Integer = 50,
                                ρ :: T = T(1e2), σ :: T = T(1e-2), adptΡ :: Bool = true, τ :: T = T(10)) where {T <: AbstractFloat}
    prm = QpsProxQpParams(numIterations, numItrConv, adptΡ, 0, ϵAbs, ϵRel, ρ, σ, τ)
    rep = QpsProxQpReport()
    _check(ccall((:qps_proxqp_solve, LIBQPS), Int32, (Ptr{Cvoid}, Ref{QpsProxQpParams}, Ref{QpsProxQpReport}), sQpProb.h, Ref(prm), rep), sQpProb.h)
    _pull_state!(sQpProb)
    return Dict{String, Real}("Converged" => rep.converged != 0, "Iterations" => rep.iterations, "ρ" => T(rep.rho), "σ" => T(rep.sigma),
                              "PrimalResidual" => T(rep.resPrim), "DualResidual" => T(rep.resDual))     # ProxQP.jl:127
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
