# lennard_jones.jl -- same exports and constructors as the reference's src/lennard_jones.jl.
export LennardJonesModel,
       LennardJonesAtom

# struct emdee_lj_model { double rc2, rs2, inv_delta2; }  -- passed BY VALUE to the C ABI.
# The reference stores these three fields as Float32 (src/lennard_jones.jl:6-9); the fp32 kernels round
# them to Float32 again, so both precisions see the reference's values.
struct LennardJonesModel
    rc²::Float64
    rs²::Float64
    δ⁻²::Float64
    function LennardJonesModel(cutoff, switch)
        0 <= switch < cutoff || throw(ArgumentError("LennardJonesModel needs 0 <= switch < cutoff"))
        new(cutoff^2, switch^2, 1/(cutoff^2 - switch^2))          # src/lennard_jones.jl:10
    end
end

LennardJonesAtom(ε, σ) = LJAtom(0.5σ, 2*sqrt(ε))                   # src/lennard_jones.jl:13

# struct emdee_lj_atom { float half_sigma, twice_sqrt_eps; }  == the reference's isbits LJAtom
struct LJAtom
    half_σ::Float32
    twice_sqrt_ε::Float32
end

# interaction(r², model, atom_i, atom_j) -> (E, minus_E′r), src/lennard_jones.jl:25-42, evaluated by the
# device pair function on a device vector of r² values.
# int32_t emdee_interaction(emdee_ctx*, int32_t n, const void *r2_dev, emdee_lj_model model,
#                           emdee_lj_atom atom_i, emdee_lj_atom atom_j, int32_t mode,
#                           void *E_dev, void *W_dev, int32_t precision);
function interaction(r²::HipArray{T,1}, model::LennardJonesModel, atom_i::LJAtom, atom_j::LJAtom; mode=0) where {T}
    E = HipArray{T,1}(undef, size(r²)); W = HipArray{T,1}(undef, size(r²))
    check(ccall((:emdee_interaction, libemdee_hip), Int32,
                (Ptr{Cvoid}, Int32, Ptr{Cvoid}, LennardJonesModel, LJAtom, LJAtom, Int32, Ptr{Cvoid}, Ptr{Cvoid}, Int32),
                context().handle, length(r²), r².ptr, model, atom_i, atom_j, mode, E.ptr, W.ptr, precision_of(T)))
    return E, W
end
