# decomposition.jl -- spatial domain decomposition (multi-GPU), bound to the emdee_dd_* entry points of
# include/emdee_hip.h.  Build-defined: the reference is single-GPU (SURVEY.md 8(b) table, 8(e)).  One Julia process
# per GPU (Distributed.jl / MPI.jl workers); rank 0 generates the communicator id and the caller hands it to the
# other ranks by its own means, e.g.   id = rank == 0 ? dd_unique_id() : nothing;  id = MPI.bcast(id, 0, comm).
export DomainDecomposition, dd_unique_id, set_atoms!, load!, owned_state!, set_overlap!

mutable struct DomainDecomposition{T}
    handle::Ptr{Cvoid}
    grid::NTuple{3,Int}
    rank::Int
end

# int32_t emdee_dd_unique_id(uint8_t out[128]);
function dd_unique_id()
    id = zeros(UInt8, 128)
    check(ccall((:emdee_dd_unique_id, libemdee_hip), Int32, (Ptr{UInt8},), id))
    return id
end

# int32_t emdee_dd_create(emdee_ctx*, const double len[3], const int32_t grid[3], int32_t rank_first, int32_t n_local,
#                         const uint8_t *unique_id, emdee_lj_model model, double skin, int32_t precision, emdee_dd **out);
# rank = this process' domain (0-based) of the grid[1] x grid[2] x grid[3] bricks, halo messages over RCCL.
function DomainDecomposition(::Type{T}, lengths, grid, rank, unique_id::Vector{UInt8}, model::LennardJonesModel; skin=0.3) where {T}
    h = Ref{Ptr{Cvoid}}(C_NULL)
    len = Float64[lengths...]; g = Int32[grid...]
    check(ccall((:emdee_dd_create, libemdee_hip), Int32,
                (Ptr{Cvoid}, Ptr{Float64}, Ptr{Int32}, Int32, Int32, Ptr{UInt8}, LennardJonesModel, Float64, Int32, Ref{Ptr{Cvoid}}),
                context().handle, len, g, rank, 1, unique_id, model, skin, precision_of(T), h))
    dd = DomainDecomposition{T}(h[], (Int(g[1]), Int(g[2]), Int(g[3])), rank)
    # int32_t emdee_dd_destroy(emdee_dd *dd);
    finalizer(d -> ccall((:emdee_dd_destroy, libemdee_hip), Int32, (Ptr{Cvoid},), d.handle), dd)
    return dd
end

# int32_t emdee_dd_set_atoms(emdee_dd*, int32_t local, int32_t n, const void *positions, const void *velocities,
#                            const emdee_lj_atom *atoms, const int64_t *gids);
# The atoms this rank holds initially (any atoms of the box: load! hands each to the brick that contains it).
set_atoms!(dd::DomainDecomposition{T}, positions::HipArray{T,2}, velocities::HipArray{T,2}, atoms::HipArray{LJAtom,1},
           gids::HipArray{Int64,1}) where {T} =
    check(ccall((:emdee_dd_set_atoms, libemdee_hip), Int32,
                (Ptr{Cvoid}, Int32, Int32, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}),
                dd.handle, 0, size(positions, 2), positions.ptr, velocities.ptr, atoms.ptr, gids.ptr))

# int32_t emdee_dd_load(emdee_dd *dd);      collective: migration, ghosts, neighbour list, forces
load!(dd::DomainDecomposition) = check(ccall((:emdee_dd_load, libemdee_hip), Int32, (Ptr{Cvoid},), dd.handle))

# int32_t emdee_dd_step(emdee_dd *dd, int32_t nsteps, double dt, int32_t rebuild_every);     collective
step!(dd::DomainDecomposition, nsteps, dt; rebuild_every=0) =
    check(ccall((:emdee_dd_step, libemdee_hip), Int32, (Ptr{Cvoid}, Int32, Float64, Int32), dd.handle, nsteps, dt, rebuild_every))

# int32_t emdee_dd_energies(emdee_dd *dd, double out[3]);    global sums, collective
function energies(dd::DomainDecomposition)
    out = zeros(Float64, 3)
    check(ccall((:emdee_dd_energies, libemdee_hip), Int32, (Ptr{Cvoid}, Ptr{Float64}), dd.handle, out))
    return (potential=out[1], kinetic=out[2], virial=out[3])
end

# int32_t emdee_dd_counts(emdee_dd *dd, int32_t local, int64_t *n_global, int32_t *n_owned, int32_t *n_ghost);
function Base.size(dd::DomainDecomposition)
    g = Ref{Int64}(0); o = Ref{Int32}(0); h = Ref{Int32}(0)
    check(ccall((:emdee_dd_counts, libemdee_hip), Int32, (Ptr{Cvoid}, Int32, Ref{Int64}, Ref{Int32}, Ref{Int32}), dd.handle, 0, g, o, h))
    return (n_global=Int(g[]), n_owned=Int(o[]), n_ghost=Int(h[]))
end

# int32_t emdee_dd_get_state(emdee_dd *dd, int32_t local, int64_t *gids, void *positions, void *velocities, void *forces);
owned_state!(dd::DomainDecomposition{T}, gids::HipArray{Int64,1}, positions::HipArray{T,2}, velocities::HipArray{T,2},
             forces::HipArray{T,2}) where {T} =
    check(ccall((:emdee_dd_get_state, libemdee_hip), Int32, (Ptr{Cvoid}, Int32, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}),
                dd.handle, 0, gids.ptr, positions.ptr, velocities.ptr, forces.ptr))

# int32_t emdee_dd_set_langevin(emdee_dd *dd, double gamma, double temperature, uint64_t seed, uint64_t first_step);
set_langevin!(dd::DomainDecomposition, gamma, temperature; seed=0, first_step=0) =
    check(ccall((:emdee_dd_set_langevin, libemdee_hip), Int32, (Ptr{Cvoid}, Float64, Float64, UInt64, UInt64),
                dd.handle, gamma, temperature, seed, first_step))

# int32_t emdee_dd_set_overlap(emdee_dd *dd, int32_t overlap);
# true (default): interior bricks overlap the halo exchange; false: exchange and one launch over all bricks in order
set_overlap!(dd::DomainDecomposition, on::Bool) =
    check(ccall((:emdee_dd_set_overlap, libemdee_hip), Int32, (Ptr{Cvoid}, Int32), dd.handle, on ? 1 : 0))

# int32_t emdee_dd_phase_times(emdee_dd *dd, double out[8]);
# (cumulative host-side times: ms inside rebuilds and their number, ms of blocking read-backs and their number, ghost share)
function phase_times(dd::DomainDecomposition)
    out = zeros(Float64, 8)
    check(ccall((:emdee_dd_phase_times, libemdee_hip), Int32, (Ptr{Cvoid}, Ptr{Float64}), dd.handle, out))
    (rebuild_ms = out[1], rebuilds = Int(out[2]), readback_ms = out[3], readbacks = Int(out[4]), ghost_fraction = out[5])
end

# int32_t emdee_dd_rebuild_stats(emdee_dd *dd, int64_t out[4]);
# (count-free rebuilds, those redone with exact counts, migrant rows per message, ghost rows the messages hold)
function rebuild_stats(dd::DomainDecomposition)
    out = zeros(Int64, 4)
    check(ccall((:emdee_dd_rebuild_stats, libemdee_hip), Int32, (Ptr{Cvoid}, Ptr{Int64}), dd.handle, out))
    (count_free = out[1], redone = out[2], migrant_rows_per_peer = out[3], ghost_rows_capacity = out[4])
end
