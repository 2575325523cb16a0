# cells.jl -- Cells / update_cells! with the reference's constructor signature (src/cells.jl:176,196).
export Cells, update_cells!

mutable struct Cells
    M::Int32
    cutoff::Float64
    N::Int
    handle::Ptr{Cvoid}
end

# Cells(r, L, cutoff; ndiv=2) -- src/cells.jl:176-194 (num_threads is a launch detail of the reference and is ignored)
# int32_t emdee_cells_create(emdee_ctx*, int32_t N, double L, double cutoff, int32_t ndiv, int32_t precision, emdee_cells **out);
# int32_t emdee_cells_M(const emdee_cells *cells, int32_t *M);
function Cells(r::HipArray{T,2}, L, cutoff; ndiv=2, num_threads=256) where {T<:Number}
    h = Ref{Ptr{Cvoid}}(C_NULL)
    check(ccall((:emdee_cells_create, libemdee_hip), Int32,
                (Ptr{Cvoid}, Int32, Float64, Float64, Int32, Int32, Ref{Ptr{Cvoid}}),
                context().handle, size(r, 2), Float64(L), Float64(cutoff), ndiv, precision_of(T), h))
    M = Ref{Int32}(0)
    check(ccall((:emdee_cells_M, libemdee_hip), Int32, (Ptr{Cvoid}, Ref{Int32}), h[], M))
    cells = Cells(M[], cutoff, size(r, 2), h[])
    # int32_t emdee_cells_destroy(emdee_cells *cells);
    finalizer(c -> ccall((:emdee_cells_destroy, libemdee_hip), Int32, (Ptr{Cvoid},), c.handle), cells)
    update_cells!(cells, r, L)
    return cells
end

# update_cells!(cells, r, L) -- src/cells.jl:196-222
# int32_t emdee_cells_update(emdee_cells *cells, const void *positions_dev);
function update_cells!(cells::Cells, r::HipArray, L)
    check(ccall((:emdee_cells_update, libemdee_hip), Int32, (Ptr{Cvoid}, Ptr{Cvoid}), cells.handle, r.ptr))
    return nothing
end

# index / population as host arrays (the reference exposes them as device fields, src/cells.jl:13-14)
# int32_t emdee_cells_arrays(const emdee_cells*, const int32_t **index, const int32_t **population,
#                            const int32_t **start, const int32_t **order);
function cell_arrays(cells::Cells)
    p = [Ref{Ptr{Int32}}(C_NULL) for _ in 1:4]
    check(ccall((:emdee_cells_arrays, libemdee_hip), Int32,
                (Ptr{Cvoid}, Ref{Ptr{Int32}}, Ref{Ptr{Int32}}, Ref{Ptr{Int32}}, Ref{Ptr{Int32}}),
                cells.handle, p[1], p[2], p[3], p[4]))
    fetch(ptr, n) = (x = Vector{Int32}(undef, n);
                     check(ccall((:emdee_memcpy_d2h, libemdee_hip), Int32, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Csize_t),
                                 context().handle, x, ptr, 4n)); x)
    return (index=fetch(p[1][], cells.N), population=fetch(p[2][], Int(cells.M)^3))
end
