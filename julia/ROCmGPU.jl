# ROCmGPU.jl -- the shim a maintainer adds to Oceananigans to route the NonhydrostaticModel hot path to
# libocnhip.so.  NOT executed in this repository (no Julia toolchain in the build image); it documents the
# reference-side binding of every entry point of include/ocnhip.h.
#
# Strategy (SURVEY.md section 8b): `ROCmGPU <: AbstractArchitecture` owns a context; the *model* is mirrored
# by an `ocn_model` handle whose device arrays are aliased by the Julia `Field`s (same parent layout), and
# the phase-level functions are overloaded so that no KernelAbstractions kernel is ever launched.

module ROCmGPUShim

using Oceananigans
using Oceananigans.Architectures: AbstractArchitecture
using OffsetArrays
import Oceananigans.Architectures: device, array_type, arch_array, device_event, architecture
import Oceananigans.TimeSteppers: time_step!, ab2_step!, rk3_substep!, store_tendencies!, update_state!,
                                  calculate_tendencies!, calculate_pressure_correction!, pressure_correct_velocities!
import Oceananigans.BoundaryConditions: fill_halo_regions!
import Oceananigans.Fields: set!

const libocnhip = get(ENV, "OCNHIP_LIB", "libocnhip.so")

check(rc, ctx=C_NULL) = rc == 0 ? nothing :
    error("libocnhip: ", unsafe_string(ccall((:ocn_last_error, libocnhip), Cstring, (Ptr{Cvoid},), ctx)))

# ---- architecture (src/Architectures.jl:53-142) ---------------------------------------------------------
mutable struct ROCmGPU <: AbstractArchitecture
    ctx :: Ptr{Cvoid}
end

function ROCmGPU(device_id::Integer = 0)
    ctx = Ref{Ptr{Cvoid}}(C_NULL)
    check(ccall((:ocn_init, libocnhip), Cint, (Cint, Ref{Ptr{Cvoid}}), device_id, ctx))
    arch = ROCmGPU(ctx[])
    finalizer(a -> ccall((:ocn_destroy, libocnhip), Cvoid, (Ptr{Cvoid},), a.ctx), arch)
    return arch
end

# A device array = library-owned memory + the parent size; host copies go through upload / download.
struct ROCArray{T, N} <: AbstractArray{T, N}
    ptr  :: Ptr{T}
    dims :: NTuple{N, Int}
end
Base.size(a::ROCArray) = a.dims
array_type(::ROCmGPU) = ROCArray
device_event(::ROCmGPU) = nothing                     # one in-order stream replaces KA events
Base.wait(::ROCmGPU, ::Nothing) = nothing

# ---- grid + model handles ----------------------------------------------------------------------------------
struct GridDesc                                        # mirrors `ocn_grid_desc` field by field
    N::NTuple{3, Int32}; H::NTuple{3, Int32}; topology::NTuple{3, Int32}
    x0::NTuple{3, Float64}; L::NTuple{3, Float64}; z_faces::Ptr{Float64}; rank::Int32; nranks::Int32
end

topo_code(::Type{Periodic}) = Int32(0); topo_code(::Type{Bounded}) = Int32(1); topo_code(::Type{Flat}) = Int32(2)

function ocn_grid(arch::ROCmGPU, grid::RectilinearGrid)
    TX, TY, TZ = topology(grid)
    zf = grid.Δzᵃᵃᶜ isa Number ? C_NULL : pointer(collect(Float64, grid.zᵃᵃᶠ[1:grid.Nz+1]))
    desc = GridDesc((grid.Nx, grid.Ny, grid.Nz), (grid.Hx, grid.Hy, grid.Hz), topo_code.((TX, TY, TZ)),
                    (grid.xᶠᵃᵃ[1], grid.yᵃᶠᵃ[1], grid.zᵃᵃᶠ[1]), (grid.Lx, grid.Ly, grid.Lz), zf, 0, 1)
    h = Ref{Ptr{Cvoid}}(C_NULL)
    check(ccall((:ocn_grid_create, libocnhip), Cint, (Ptr{Cvoid}, Ref{GridDesc}, Ref{Ptr{Cvoid}}), arch.ctx, desc, h), arch.ctx)
    return h[]
end

# `ocn_model_desc` (include/ocnhip.h; OCN_ABI_VERSION 5), field by field.  isbits, so it crosses ccall by reference.
const MAXTR = 8
struct BC; kind::Int32; value::Float64; array::Ptr{Float64}; end                       # ocn_bc
struct ModelDesc
    advection::Int32; stepper::Int32; chi::Float64; n_tracers::Int32; closure::Int32
    nu::Float64; kappa::NTuple{MAXTR, Float64}
    amd_Cnu::Float64; amd_Ckappa::NTuple{MAXTR, Float64}; amd_Cb::Float64; amd_has_Cb::Int32
    coriolis_fplane::Int32; f::Float64
    buoyancy::Int32; b_index::Int32; T_index::Int32; S_index::Int32; g::Float64; alpha::Float64; beta::Float64
    bcs::NTuple{3 + MAXTR, NTuple{6, BC}}                                               # [u, v, w, tracers...][west .. top]
    nu_bcs::NTuple{6, BC}; kappa_bcs::NTuple{MAXTR, NTuple{6, BC}}                      # AMD diffusivity fields (model.diffusivity_fields)
end

adv_code(::Nothing) = Int32(0); adv_code(::CenteredSecondOrder) = Int32(1); adv_code(::CenteredFourthOrder) = Int32(2)
adv_code(::UpwindBiasedFifthOrder) = Int32(3); adv_code(a::WENO5) = a isa WENO5{<:Any, <:Any, <:Any, <:Any, <:Any, true} ? Int32(4) : Int32(5)  # zweno / JS
adv_code(::UpwindBiasedFirstOrder) = Int32(6); adv_code(::UpwindBiasedThirdOrder) = Int32(7)

# one side of one field: BoundaryCondition{Flux | Value | Gradient | Open | Periodic, Nothing | Number | Array}
function bc_of(bc, keep)
    cls = bc.classification
    kind = cls isa Periodic ? 1 : cls isa Flux ? (bc.condition === nothing ? 2 : 3) : cls isa Value ? 4 : cls isa Gradient ? 5 :
           cls isa Open ? 6 : 0
    c = bc.condition
    c isa Function && error("function boundary conditions cannot cross the C ABI")     # ContinuousBoundaryFunction etc.
    c isa AbstractArray && (a = collect(Float64, c); push!(keep, a); return BC(kind, 0.0, pointer(a)))
    return BC(kind, c === nothing ? 0.0 : Float64(c), C_NULL)
end

function ocn_model(arch::ROCmGPU, gridh, m)                                             # m: the fields NonhydrostaticModel(...) assembled
    isempty(m.forcing) && m.stokes_drift === nothing && isempty(m.background_fields) ||
        error("forcings, Stokes drift and background fields are Julia closures: outside the C ABI")
    names = keys(m.tracers); nt = length(names)
    pad(t) = ntuple(i -> i <= length(t) ? Float64(t[i]) : 0.0, MAXTR)
    cl = m.closure
    closure, nu, kap, Cnu, Ck, Cb, hasCb = Int32(0), 0.0, pad(()), 0.0, pad(()), 0.0, Int32(0)
    if cl isa ScalarDiffusivity
        closure, nu, kap = Int32(1), cl.ν, pad(values(cl.κ))
    elseif cl isa AnisotropicMinimumDissipation
        closure, Cnu, Ck = Int32(2), cl.Cν, pad(values(cl.Cκ))
        cl.Cb === nothing || ((Cb, hasCb) = (Float64(cl.Cb), Int32(1)))
    elseif cl !== nothing
        error("closure $(typeof(cl)) is outside the path")
    end
    idx(n) = Int32(something(findfirst(==(n), names), 0) - 1)
    b = m.buoyancy === nothing ? nothing : m.buoyancy.model
    buoy, g, α, β = b === nothing ? (Int32(0), 0.0, 0.0, 0.0) : b isa BuoyancyTracer ? (Int32(1), 0.0, 0.0, 0.0) :
                    (Int32(2), b.gravitational_acceleration, b.equation_of_state.thermal_expansion, b.equation_of_state.haline_contraction)
    keep = Any[]                                                                        # host arrays stay alive across the call
    sides(f) = (bcs = f.boundary_conditions; ntuple(s -> bc_of((bcs.west, bcs.east, bcs.south, bcs.north, bcs.bottom, bcs.top)[s], keep), 6))
    none = ntuple(_ -> BC(0, 0.0, C_NULL), 6)
    fields = (m.velocities.u, m.velocities.v, m.velocities.w, values(m.tracers)...)
    desc = ModelDesc(adv_code(m.advection), m.timestepper isa RungeKutta3TimeStepper ? 1 : 0,
                     m.timestepper isa QuasiAdamsBashforth2TimeStepper ? m.timestepper.χ : 0.1, nt, closure, nu, kap, Cnu, Ck, Cb, hasCb,
                     m.coriolis === nothing ? 0 : 1, m.coriolis === nothing ? 0.0 : m.coriolis.f,
                     buoy, idx(:b), idx(:T), idx(:S), g, α, β,
                     ntuple(i -> i <= length(fields) ? sides(fields[i]) : none, 3 + MAXTR),
                     closure == 2 ? sides(m.diffusivity_fields.νₑ) : none,
                     ntuple(i -> (closure == 2 && i <= nt) ? sides(m.diffusivity_fields.κₑ[i]) : none, MAXTR))
    h = Ref{Ptr{Cvoid}}(C_NULL)
    GC.@preserve keep check(ccall((:ocn_model_create, libocnhip), Cint, (Ptr{Cvoid}, Ref{ModelDesc}, Ref{Ptr{Cvoid}}), gridh, desc, h), arch.ctx)
    return h[]
end

# Stand-alone fields -- Field{LX, LY, LZ}(grid), CenterField(grid), zeros(FT, ::ROCmGPU, N...) (Fields/field.jl:16-30,
# Grids/zeros.jl:7): ocn_field_create gives a zero-filled parent array with the layout of the model's own fields.
loc_code(::Type{Center}) = Int32(0); loc_code(::Type{Face}) = Int32(1)
function new_field_data(gridh, grid, loc, ctx)
    h = Ref{Ptr{Cvoid}}(C_NULL)
    check(ccall((:ocn_field_create, libocnhip), Cint, (Ptr{Cvoid}, Cint, Cint, Cint, Ref{Ptr{Cvoid}}), gridh, loc_code.(loc)..., h), ctx)
    p = ccall((:ocn_field_parent_ptr, libocnhip), Ptr{Float64}, (Ptr{Cvoid},), h[])
    T = Oceananigans.Grids.total_size(loc, grid)      # dense on (Periodic, Periodic, *) grids; ocn_field_parent_layout otherwise, as in alias_field_data
    return Oceananigans.Grids.offset_data(unsafe_wrap(ROCArray, p, T), grid, loc), h[]   # keep h: ocn_field_destroy in the finalizer
end

# Field data alias the library's parent arrays (same layout as Grids/new_data.jl:33-61).  With walls or slices in
# x / y the allocation is pitched: ocn_field_layout gives element strides and the origin of the logical parent.
# u, v, w, both pressures, nu_e and kappa_e keep their device pointers for the life of the model (alias once, at model
# construction); G^n / G^- and, on the tiled kernels' paths, the tracers are double buffered: call this again after every
# time_step! for those fields (include/ocnhip.h, "pointer stability"; tests/test_model_contracts.py).
function alias_field_data(model_handle, field_id, grid, loc)
    p = ccall((:ocn_field_device_ptr, libocnhip), Ptr{Float64}, (Ptr{Cvoid}, Cint), model_handle, field_id)
    st = zeros(Int64, 3); org = Ref(Int64(0))
    ccall((:ocn_field_layout, libocnhip), Cint, (Ptr{Cvoid}, Cint, Ptr{Int64}, Ref{Int64}), model_handle, field_id, st, org)
    T = Oceananigans.Grids.total_size(loc, grid)
    Px, Py = st[2], st[3] ÷ st[2]
    whole = unsafe_wrap(ROCArray, p, (Px, Py, T[3]))
    ox, oy = org[] % Px, org[] ÷ Px
    parent = (Px, Py) == T[1:2] ? whole : view(whole, ox .+ (1:T[1]), oy .+ (1:T[2]), :)
    return Oceananigans.Grids.offset_data(parent, grid, loc)
end

# ---- phase-level overloads (each is ONE ccall) ---------------------------------------------------------------
const RM = NonhydrostaticModel{<:Any, <:Any, <:ROCmGPU}   # models living on ROCmGPU (handle kept in model.auxiliary_fields.ocn)
handle(m) = m.auxiliary_fields.ocn

# TimeSteppers/quasi_adams_bashforth_2.jl:70-104 and runge_kutta_3.jl:81-152 -- preferred, coarsest overload
function time_step!(model::RM, Δt; euler=false)
    check(ccall((:ocn_time_step, libocnhip), Cint, (Ptr{Cvoid}, Cdouble, Cint), handle(model), Δt, euler), model.architecture.ctx)
    # clock mirrors ocn_clock (TimeSteppers/clock.jl:48-60)
    t = Ref(0.0); it = Ref(Int64(0)); st = Ref(Int32(0))
    ccall((:ocn_clock, libocnhip), Cint, (Ptr{Cvoid}, Ref{Float64}, Ref{Int64}, Ref{Int32}), handle(model), t, it, st)
    model.clock.time, model.clock.iteration, model.clock.stage = t[], it[], st[]
    return nothing
end

# finer-grained overloads, for callers that drive the phases themselves
update_state!(model::RM) = check(ccall((:ocn_update_state, libocnhip), Cint, (Ptr{Cvoid},), handle(model)))                        # update_nonhydrostatic_model_state.jl:14
calculate_tendencies!(model::RM) = check(ccall((:ocn_compute_tendencies, libocnhip), Cint, (Ptr{Cvoid},), handle(model)))           # calculate_nonhydrostatic_tendencies.jl:12
ab2_step!(model::RM, Δt, χ) = check(ccall((:ocn_ab2_step, libocnhip), Cint, (Ptr{Cvoid}, Cdouble, Cdouble), handle(model), Δt, χ))  # quasi_adams_bashforth_2.jl:116
rk3_substep!(model::RM, Δt, γ, ζ) = check(ccall((:ocn_rk3_substep, libocnhip), Cint, (Ptr{Cvoid}, Cdouble, Cdouble, Cdouble, Cint),
                                                handle(model), Δt, γ, something(ζ, 0.0), ζ !== nothing))                            # runge_kutta_3.jl:161
store_tendencies!(model::RM) = check(ccall((:ocn_store_tendencies, libocnhip), Cint, (Ptr{Cvoid},), handle(model)))                 # store_tendencies.jl:14
calculate_pressure_correction!(model::RM, Δt) = check(ccall((:ocn_pressure_correction, libocnhip), Cint, (Ptr{Cvoid}, Cdouble), handle(model), Δt))           # pressure_correction.jl:10
pressure_correct_velocities!(model::RM, Δt) = check(ccall((:ocn_pressure_correct_velocities, libocnhip), Cint, (Ptr{Cvoid}, Cdouble), handle(model), Δt))     # pressure_correction.jl:43

# set!(model; ...) : host arrays -> ocn_field_set_interior per field, then the epilogue (set_nonhydrostatic_model.jl:45-58)
function set!(model::RM; enforce_incompressibility=true, kwargs...)
    for (name, value) in kwargs
        host = Array{Float64}(undef, size(getproperty(merge(model.velocities, model.tracers), name)))
        set!(CPUField(name, model), value); host .= interior(CPUField(name, model))     # stage on the CPU exactly like Fields/set!.jl:21-27
        check(ccall((:ocn_field_set_interior, libocnhip), Cint, (Ptr{Cvoid}, Cint, Ptr{Float64}), handle(model), field_id(model, name), host))
    end
    check(ccall((:ocn_set_epilogue, libocnhip), Cint, (Ptr{Cvoid}, Cint), handle(model), enforce_incompressibility))
end

# Output / checkpoint fetch (OutputWriters/fetch_output.jl:26, checkpointer.jl:158-180): parent arrays incl. G^n, G^-
function arch_array(::CPU, a::OffsetArray{T, 3, <:ROCArray}) where T
    host = Array{T}(undef, size(parent(a)))
    # the owning model + field id travel with the wrapper in the real shim; shown here for one field:
    # check(ccall((:ocn_field_download, libocnhip), Cint, (Ptr{Cvoid}, Cint, Ptr{Float64}), model_handle, field_id, host))
    return OffsetArray(host, a.offsets...)
end

# Which kernels serve a model (and why not the fastest ones): ocn_model_path(handle, buf, n) -> String, for `show(model)`.
kernel_path(model::RM) = (buf = Vector{UInt8}(undef, 256);
                          check(ccall((:ocn_model_path, libocnhip), Cint, (Ptr{Cvoid}, Ptr{UInt8}, Csize_t), handle(model), buf, 256));
                          unsafe_string(pointer(buf)))

# Checkpoint / restart (OutputWriters/checkpointer.jl:158-265): download the parent arrays of u, v, w, tracers, G^n, G^- and
# the clock; restore with ocn_field_upload + ocn_set_clock(time, iteration, previous_dt) + update_state!.  Continuing from
# the restored state is bit-identical to the uninterrupted run (tests/test_model_contracts.py, all four stepper / topology pairs).

# Distributed: MultiArch(ROCmGPU(); ranks=(1, 1, R)) -> ocn_comm_init(ctx, rank, R, id) with the 128-byte id of
# ocn_comm_unique_id broadcast over MPI (Distributed/multi_architectures.jl:20-47); everything else is unchanged.
# Before the blocking ocn_comm_init every rank calls ocn_comm_probe(ctx, rank, R, id0, timeout_s) with an id of its own and the
# ranks MPI.Allreduce(min) the return codes: a rank whose RCCL cannot start then stops everybody instead of hanging them.
#
# HydrostaticFreeSurfaceModel, first slice (Models/HydrostaticFreeSurfaceModels/split_explicit_free_surface*.jl):
#   struct ROCmSplitExplicit; grid::Ptr{Cvoid}; sefs::Ptr{Cvoid}; end                      # ocn_hgrid_create + ocn_sefs_create
#   FreeSurface(fs::SplitExplicitFreeSurface, velocities, grid) on a ROCmGPU grid -> fields aliased from ocn_sefs_field(sefs, 0:10)
#   ab2_step_free_surface!(fs, model, Δt, χ, _) = check(ccall((:ocn_sefs_step, libocnhip), Cint,
#       (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Cdouble, Cdouble), fs.sefs, Gⁿ.u, Gⁿ.v, G⁻.u, G⁻.v, Δt, χ))
#   barotropic_split_explicit_corrector!(u, v, fs, grid) = check(ccall((:ocn_sefs_corrector, libocnhip), Cint,
#       (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}), fs.sefs, hfield(u), hfield(v)))
#
# Launch-bound models (config 1) are replayed from hipGraphs inside ocn_time_step; ocn_model_graph_replays(handle, n, active)
# reports it.  The library reports OCN_ABI_VERSION through ocn_abi_version(): __init__ compares it with 5.
function __init__()
    v = ccall((:ocn_abi_version, libocnhip), Cint, ())
    v == 5 || error("libocnhip reports ABI version $v; this shim is written for 5")
end

end # module
