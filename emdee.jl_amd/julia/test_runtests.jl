# The reference's GPU test (test/runtests.jl:19-42,58) against this binding.  Not executed here (no Julia
# in the image); tests/test_gpu_parity.py::test_compute_nonbonded_reference_test is the executed twin.
using Test
include("src/EmDee.jl")
using .EmDee

function read_xyz(file)                      # stands in for Chemfiles (test/runtests.jl:20-21)
    lines = readlines(file)
    N = parse(Int, lines[1])
    xyz = Matrix{Float64}(undef, 3, N)
    for i in 1:N
        xyz[:, i] = parse.(Float64, split(lines[i+2])[2:4])
    end
    return xyz
end

function test_compute_nonbonded(xyz_file, L, cutoff, switch)
    xyz_data = read_xyz(xyz_file)
    positions = EmDee.cu(Float32.(xyz_data))
    N = size(xyz_data, 2)
    model = LennardJonesModel(cutoff, switch)
    atoms = EmDee.cu(fill(LennardJonesAtom(1, 1), N))

    forces_ref = EmDee.zeros(Float32, 3, N)
    energies_ref = EmDee.zeros(Float32, N)
    virials_ref = EmDee.zeros(Float32, N)
    naively_compute_nonbonded!(forces_ref, energies_ref, virials_ref, positions, L, model, atoms)

    tiles = nonbonded_computation_tiles(N, all_pairs=true)
    forces = EmDee.zeros(Float32, 3, N)
    energies = EmDee.zeros(Float32, N)
    virials = EmDee.zeros(Float32, N)
    compute_nonbonded!(forces, energies, virials, positions, L,
                       tiles, model, atoms, Val(FORCES | ENERGIES | VIRIALS))

    return maximum(abs.(Array(forces) .- Array(forces_ref))) < 1.0f-4 &&
           maximum(abs.(Array(energies) .- Array(energies_ref))) < 1.0f-4 &&
           maximum(abs.(Array(virials) .- Array(virials_ref))) < 1.0f-4
end

@testset "EmDee[GPU]" begin
    @test test_compute_nonbonded(joinpath(@__DIR__, "..", "..", "tests", "golden", "lj_sample.xyz"), 10, 3, 2.5)
end
