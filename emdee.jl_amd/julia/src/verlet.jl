# verlet.jl -- velocity-Verlet on the device.  Build-defined: the reference has no integrator
# (SURVEY.md 8a row a16).  API in EmDee's style: a constructor and `!` mutators.
export VelocityVerlet, step!, energies, set_langevin!, set_langevin_ids!

mutable struct VelocityVerlet{T}
    handle::Ptr{Cvoid}
    N::Int
end

# int32_t emdee_md_create(emdee_ctx*, const double lo[3], const double len[3], const int32_t periodic[3],
#                         emdee_lj_model model, double skin, int32_t precision, emdee_md **out);
# int32_t emdee_md_set_state(emdee_md*, int32_t n_owned, int32_t n_ghost, const void *positions,
#                            const void *velocities, const emdee_lj_atom *atoms, const void *inv_mass);
function VelocityVerlet(positions::HipArray{T,2}, velocities::HipArray{T,2}, L, model::LennardJonesModel,
                        atoms::HipArray{LJAtom,1}; skin=0.3) where {T}
    h = Ref{Ptr{Cvoid}}(C_NULL)
    lo = Float64[0, 0, 0]; len = Float64[L, L, L]; per = Int32[1, 1, 1]
    check(ccall((:emdee_md_create, libemdee_hip), Int32,
                (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ptr{Int32}, LennardJonesModel, Float64, Int32, Ref{Ptr{Cvoid}}),
                context().handle, lo, len, per, model, skin, precision_of(T), h))
    md = VelocityVerlet{T}(h[], size(positions, 2))
    # int32_t emdee_md_destroy(emdee_md *md);
    finalizer(m -> ccall((:emdee_md_destroy, libemdee_hip), Int32, (Ptr{Cvoid},), m.handle), md)
    check(ccall((:emdee_md_set_state, libemdee_hip), Int32,
                (Ptr{Cvoid}, Int32, Int32, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}),
                md.handle, md.N, 0, positions.ptr, velocities.ptr, atoms.ptr, C_NULL))
    return md
end

# int32_t emdee_md_step(emdee_md *md, int32_t nsteps, double dt, int32_t rebuild_every);
step!(md::VelocityVerlet, nsteps, dt; rebuild_every=0) =
    check(ccall((:emdee_md_step, libemdee_hip), Int32, (Ptr{Cvoid}, Int32, Float64, Int32), md.handle, nsteps, dt, rebuild_every))

# int32_t emdee_md_set_langevin(emdee_md *md, double gamma, double temperature, uint64_t seed, uint64_t first_step);
# gamma > 0: every later step does v = c1 v + c2 sqrt(T/m) xi(seed, step, atom) between its kick and its drift.
set_langevin!(md::VelocityVerlet, gamma, temperature; seed=0, first_step=0) =
    check(ccall((:emdee_md_set_langevin, libemdee_hip), Int32, (Ptr{Cvoid}, Float64, Float64, UInt64, UInt64),
                md.handle, gamma, temperature, seed, first_step))

# int32_t emdee_md_set_exclusions(emdee_md *md, const int32_t *pairs_dev, int32_t n_pairs);
# int32_t emdee_md_set_pairs14(emdee_md *md, const int32_t *pairs_dev, int32_t n_pairs, double lj14scale);
# pairs: 2 x n device matrix of 0-based atom indices (the hooks of src/modelling.jl:197-200: bonded neighbours contribute
# nothing, 1-4 pairs lj14scale times their pair terms); after the state is loaded; `nothing` clears the table.
set_exclusions!(md::VelocityVerlet, pairs::Union{Nothing,HipArray{Int32,2}}) =
    check(ccall((:emdee_md_set_exclusions, libemdee_hip), Int32, (Ptr{Cvoid}, Ptr{Cvoid}, Int32), md.handle,
                pairs === nothing ? C_NULL : pairs.ptr, pairs === nothing ? 0 : size(pairs, 2)))
set_pairs14!(md::VelocityVerlet, pairs::Union{Nothing,HipArray{Int32,2}}, lj14scale) =
    check(ccall((:emdee_md_set_pairs14, libemdee_hip), Int32, (Ptr{Cvoid}, Ptr{Cvoid}, Int32, Float64), md.handle,
                pairs === nothing ? C_NULL : pairs.ptr, pairs === nothing ? 0 : size(pairs, 2), Float64(lj14scale)))

# int32_t emdee_md_set_langevin_ids(emdee_md *md, const int64_t *ids_dev);
# Atom ids keying the thermostat's noise (device Int64 vector in caller order); `nothing` = the caller index.
set_langevin_ids!(md::VelocityVerlet, ids::Union{Nothing,HipArray{Int64,1}}) =
    check(ccall((:emdee_md_set_langevin_ids, libemdee_hip), Int32, (Ptr{Cvoid}, Ptr{Cvoid}), md.handle,
                ids === nothing ? C_NULL : ids.ptr))

# int32_t emdee_md_energies(emdee_md *md, double out[3]);   -> (potential, kinetic, virial sum)
function energies(md::VelocityVerlet)
    out = zeros(Float64, 3)
    check(ccall((:emdee_md_energies, libemdee_hip), Int32, (Ptr{Cvoid}, Ptr{Float64}), md.handle, out))
    return (potential=out[1], kinetic=out[2], virial=out[3])
end

# int32_t emdee_md_get_state(emdee_md*, void *positions, void *velocities, void *forces, void *energies, void *virials);
function state!(md::VelocityVerlet{T}, positions::HipArray{T,2}, velocities::HipArray{T,2}, forces::HipArray{T,2}) where {T}
    check(ccall((:emdee_md_get_state, libemdee_hip), Int32,
                (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}),
                md.handle, positions.ptr, velocities.ptr, forces.ptr, C_NULL, C_NULL))
    return nothing
end
