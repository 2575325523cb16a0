# hiparray.jl -- minimal device array + context; stands in for CuArray / CUDA.cu / CUDA.zeros /
# Array(dev) (src/nonbonded.jl:25,123,151-153; test/runtests.jl:22-35).
export HipArray, cu, context

mutable struct Context
    handle::Ptr{Cvoid}
end

const CONTEXT = Ref{Union{Nothing,Context}}(nothing)

# int32_t emdee_ctx_create(int32_t device_id, void *stream, emdee_ctx **out);
function context(device::Integer=0)
    if CONTEXT[] === nothing
        h = Ref{Ptr{Cvoid}}(C_NULL)
        check(ccall((:emdee_ctx_create, libemdee_hip), Int32, (Int32, Ptr{Cvoid}, Ref{Ptr{Cvoid}}), device, C_NULL, h))
        ctx = Context(h[])
        # int32_t emdee_ctx_destroy(emdee_ctx *ctx);
        finalizer(c -> ccall((:emdee_ctx_destroy, libemdee_hip), Int32, (Ptr{Cvoid},), c.handle), ctx)
        CONTEXT[] = ctx
    end
    return CONTEXT[]
end

# int32_t emdee_sync(emdee_ctx *ctx);
synchronize() = check(ccall((:emdee_sync, libemdee_hip), Int32, (Ptr{Cvoid},), context().handle))

mutable struct HipArray{T,N} <: AbstractArray{T,N}
    ptr::Ptr{T}
    dims::NTuple{N,Int}
    function HipArray{T,N}(::UndefInitializer, dims::NTuple{N,Int}) where {T,N}
        isbitstype(T) || error("HipArray needs an isbits element type")
        p = Ref{Ptr{Cvoid}}(C_NULL)
        # int32_t emdee_malloc(emdee_ctx *ctx, size_t nbytes, void **dev);
        check(ccall((:emdee_malloc, libemdee_hip), Int32, (Ptr{Cvoid}, Csize_t, Ref{Ptr{Cvoid}}),
                    context().handle, prod(dims)*sizeof(T), p))
        a = new{T,N}(convert(Ptr{T}, p[]), dims)
        # int32_t emdee_free(emdee_ctx *ctx, void *dev);
        finalizer(x -> ccall((:emdee_free, libemdee_hip), Int32, (Ptr{Cvoid}, Ptr{Cvoid}), context().handle, x.ptr), a)
        return a
    end
end

Base.size(a::HipArray) = a.dims
Base.pointer(a::HipArray) = a.ptr
Base.getindex(::HipArray, i...) = error("scalar indexing of a device array is disallowed (cf. CUDA.allowscalar(false), src/nonbonded.jl:10)")

# CUDA.cu(x): host -> device.  Unlike CUDA.cu it keeps Float64 as Float64 (north-star precision);
# pass Float32 data for the reference's own precision.
# int32_t emdee_memcpy_h2d(emdee_ctx *ctx, void *dev, const void *host, size_t nbytes);
function cu(x::Array{T,N}) where {T,N}
    a = HipArray{T,N}(undef, size(x))
    check(ccall((:emdee_memcpy_h2d, libemdee_hip), Int32, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Csize_t),
                context().handle, a.ptr, x, sizeof(x)))
    return a
end

# Array(dev): device -> host (blocking, like the reference's implicit synchronisation, test/runtests.jl:39)
# int32_t emdee_memcpy_d2h(emdee_ctx *ctx, void *host, const void *dev, size_t nbytes);
function Base.Array(a::HipArray{T,N}) where {T,N}
    x = Array{T,N}(undef, a.dims)
    check(ccall((:emdee_memcpy_d2h, libemdee_hip), Int32, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Csize_t),
                context().handle, x, a.ptr, sizeof(x)))
    return x
end

# CUDA.zeros(T, dims...)
# int32_t emdee_memset(emdee_ctx *ctx, void *dev, int32_t byte, size_t nbytes);
function zeros(::Type{T}, dims::Int...) where {T}
    a = HipArray{T,length(dims)}(undef, dims)
    check(ccall((:emdee_memset, libemdee_hip), Int32, (Ptr{Cvoid}, Ptr{Cvoid}, Int32, Csize_t),
                context().handle, a.ptr, 0, prod(dims)*sizeof(T)))
    return a
end

precision_of(::Type{Float32}) = Int32(4)    # EMDEE_F32
precision_of(::Type{Float64}) = Int32(8)    # EMDEE_F64
