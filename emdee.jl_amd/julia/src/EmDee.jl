# EmDee.jl -- module EmDee, same name as the reference package (src/EmDee.jl:1): drop-in for the hot-path half of EmDee.jl (src/EmDee.jl:3-5 includes vec3.jl,
# lennard_jones.jl, nonbonded.jl).  No CUDA.jl, no AMDGPU.jl kernel DSL: every kernel is hand-written
# HIP in libemdee_hip.so, reached with ccall, following the reference's own ccall precedent
# (src/molecular_graphs.jl:73-80).
module EmDee

import Libdl

const libemdee_hip = get(ENV, "EMDEE_HIP_LIB", "libemdee_hip.so")

struct EmDeeError <: Exception
    code::Int32
    msg::String
end

# every entry point returns an int32 status; the message is per calling thread
function check(status::Int32)
    status == 0 && return nothing
    msg = unsafe_string(ccall((:emdee_last_error, libemdee_hip), Cstring, ()))
    throw(EmDeeError(status, msg))
end

include("hiparray.jl")
include("lennard_jones.jl")
include("nonbonded.jl")
include("cells.jl")
include("verlet.jl")
include("decomposition.jl")

end
