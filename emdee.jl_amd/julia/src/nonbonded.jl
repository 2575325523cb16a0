# nonbonded.jl -- same exports, names and argument order as the reference's src/nonbonded.jl.
export FORCES,
       ENERGIES,
       VIRIALS,
       nonbonded_computation_tiles,
       compute_nonbonded!,
       naively_compute_nonbonded!

const FORCES = 1 << 0          # src/nonbonded.jl:12-14
const ENERGIES = 1 << 1
const VIRIALS = 1 << 2

const WAVESIZE = 64            # the reference's WARPSIZE = 32 (src/nonbonded.jl:16); a CDNA4 wavefront has 64 lanes

const LITERAL = Int32(0)       # reference formula for every pair (full LJ beyond rc, SURVEY Q1)
const CUTOFF = Int32(1)        # pairs with r² >= rc² contribute nothing

# What nonbonded_computation_tiles(N) returns (src/nonbonded.jl:18-26).  The reference returns the
# n(n+1)/2 tile pairs of the all-pairs matrix; here it is the O(N) neighbour-list workspace that
# compute_nonbonded! iterates over instead, created lazily for the element type of the first call.
mutable struct NeighborTiles
    N::Int
    skin::Float64
    handle::Ptr{Cvoid}
    precision::Int32
end

struct AllPairsTiles           # the reference's own all-pairs tile semantics, 64 x 64 tiles
    N::Int
    mode::Int32
end

nonbonded_computation_tiles(N; skin=0.3, all_pairs=false, mode=LITERAL) =
    all_pairs ? AllPairsTiles(N, mode) : NeighborTiles(N, skin, C_NULL, 0)

# int32_t emdee_nbr_create(emdee_ctx*, int32_t N, double skin, int32_t precision, emdee_nbr **out);
# int32_t emdee_nbr_destroy(emdee_nbr *nbr);
function handle!(tiles::NeighborTiles, precision::Int32)
    if tiles.handle == C_NULL || tiles.precision != precision
        tiles.handle != C_NULL && ccall((:emdee_nbr_destroy, libemdee_hip), Int32, (Ptr{Cvoid},), tiles.handle)
        h = Ref{Ptr{Cvoid}}(C_NULL)
        check(ccall((:emdee_nbr_create, libemdee_hip), Int32, (Ptr{Cvoid}, Int32, Float64, Int32, Ref{Ptr{Cvoid}}),
                    context().handle, tiles.N, tiles.skin, precision, h))
        tiles.handle, tiles.precision = h[], precision
        finalizer(t -> ccall((:emdee_nbr_destroy, libemdee_hip), Int32, (Ptr{Cvoid},), t.handle), tiles)
    end
    return tiles.handle
end

# int32_t emdee_nbr_set_exclusions(emdee_nbr *nbr, const int32_t *pairs_dev, int32_t n_pairs);
# int32_t emdee_nbr_set_pairs14(emdee_nbr *nbr, const int32_t *pairs_dev, int32_t n_pairs, double lj14scale);
# Exclusions and scaled 1-4 pairs of a molecular model (build-defined: the reference parses lj14scale, src/modelling.jl:197-200,
# and its hot path never uses it): 2 x n device matrices of 0-based atom indices.  The handle must exist (handle!(tiles, precision)).
set_exclusions!(tiles::NeighborTiles, pairs::Union{Nothing,HipArray{Int32,2}}; precision::Int32=Int32(8)) =
    check(ccall((:emdee_nbr_set_exclusions, libemdee_hip), Int32, (Ptr{Cvoid}, Ptr{Cvoid}, Int32), handle!(tiles, precision),
                pairs === nothing ? C_NULL : pairs.ptr, pairs === nothing ? 0 : size(pairs, 2)))
set_pairs14!(tiles::NeighborTiles, pairs::Union{Nothing,HipArray{Int32,2}}, lj14scale; precision::Int32=Int32(8)) =
    check(ccall((:emdee_nbr_set_pairs14, libemdee_hip), Int32, (Ptr{Cvoid}, Ptr{Cvoid}, Int32, Float64), handle!(tiles, precision),
                pairs === nothing ? C_NULL : pairs.ptr, pairs === nothing ? 0 : size(pairs, 2), Float64(lj14scale)))

# compute_nonbonded!(forces, energies, virials, positions, L, tiles, model, atoms, Val(bitmask))
# -- src/nonbonded.jl:109-120.  positions/forces are 3xN device matrices, energies/virials length N,
# atoms a device vector of LJAtom.  Selected outputs are overwritten; asynchronous like the reference.
# int32_t emdee_compute_nonbonded(emdee_ctx*, void *forces, void *energies, void *virials,
#                                 const void *positions, double L, emdee_nbr *nbr, emdee_lj_model model,
#                                 const emdee_lj_atom *atoms, int32_t bitmask, int32_t precision);
function compute_nonbonded!(forces, energies, virials, positions::HipArray{T,2}, L,
                            tiles::NeighborTiles, model::LennardJonesModel, atoms::HipArray{LJAtom,1},
                            ::Val{bitmask}) where {T, bitmask}
    size(positions, 2) == tiles.N || throw(DimensionMismatch("tiles were built for N = $(tiles.N)"))
    check(ccall((:emdee_compute_nonbonded, libemdee_hip), Int32,
                (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Float64, Ptr{Cvoid}, LennardJonesModel,
                 Ptr{Cvoid}, Int32, Int32),
                context().handle, forces.ptr, energies.ptr, virials.ptr, positions.ptr, Float64(L),
                handle!(tiles, precision_of(T)), model, atoms.ptr, bitmask, precision_of(T)))
    return nothing
end

# int32_t emdee_compute_nonbonded_tiles(emdee_ctx*, void *forces, void *energies, void *virials,
#                                       const void *positions, double L, int32_t N, emdee_lj_model model,
#                                       const emdee_lj_atom *atoms, int32_t bitmask, int32_t mode, int32_t precision);
function compute_nonbonded!(forces, energies, virials, positions::HipArray{T,2}, L,
                            tiles::AllPairsTiles, model::LennardJonesModel, atoms::HipArray{LJAtom,1},
                            ::Val{bitmask}) where {T, bitmask}
    check(ccall((:emdee_compute_nonbonded_tiles, libemdee_hip), Int32,
                (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Float64, Int32, LennardJonesModel,
                 Ptr{Cvoid}, Int32, Int32, Int32),
                context().handle, forces.ptr, energies.ptr, virials.ptr, positions.ptr, Float64(L), tiles.N,
                model, atoms.ptr, bitmask, tiles.mode, precision_of(T)))
    return nothing
end

# naively_compute_nonbonded!(forces, energies, virials, positions, L, model, atoms) -- src/nonbonded.jl:122-155.
# The reference runs this double loop on the host; here it is a device kernel (one thread per atom).
# int32_t emdee_compute_nonbonded_naive(emdee_ctx*, void *forces, void *energies, void *virials,
#                                       const void *positions, double L, int32_t N, emdee_lj_model model,
#                                       const emdee_lj_atom *atoms, int32_t mode, int32_t precision);
function naively_compute_nonbonded!(forces, energies, virials, positions::HipArray{T,2}, L, model, atoms; mode=LITERAL) where {T}
    check(ccall((:emdee_compute_nonbonded_naive, libemdee_hip), Int32,
                (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Float64, Int32, LennardJonesModel,
                 Ptr{Cvoid}, Int32, Int32),
                context().handle, forces.ptr, energies.ptr, virials.ptr, positions.ptr, Float64(L),
                size(positions, 2), model, atoms.ptr, mode, precision_of(T)))
    return nothing
end
