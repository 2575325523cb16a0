// allpairs.hpp -- O(N^2) device kernels with the reference operator's exact pair set.
//
//  k_tiles  compute_tile! (src/nonbonded.jl:44-107) re-designed for wave64: the interaction matrix
//           is cut into 64x64 tiles; lane t of a wavefront owns atom I = 64 bI + t and holds
//           atom J = 64 bJ + t; 64 lane rotations (ds_bpermute instead of the reference's
//           32-wide shfl_sync) bring every J atom past every I atom.  Owner-computes: a block
//           owns one I tile and its 4 waves sweep the J tiles, so there is no j-side
//           accumulation, no float atomics and no pre-zeroing (the reference needs all three,
//           src/nonbonded.jl:88-104,112-114), and the result is bitwise reproducible.  Any N
//           (the reference needs N % 32 == 0).
//  k_naive  naively_compute_nonbonded! (src/nonbonded.jl:122-155): the plain double loop, one
//           thread per atom i over all j != i.
//
// Both follow the reference's coordinates: scaled positions s = r / L (:52-61,124), minimum
// image on s (:40,70,136), r_ij = L * image.  MODE = EMDEE_LITERAL evaluates the reference
// formula for every pair (g = 1 beyond rc, Q1); EMDEE_CUTOFF drops r2 >= rc2.
#pragma once

#include "kernels.hpp"

namespace emdee {

constexpr int TILE = WAVE;
constexpr int TILE_BLOCK = 256;
constexpr int TILE_WAVES = TILE_BLOCK / WAVE;

template <typename real, int MODE>
__device__ __forceinline__ void pair_terms(real dsx, real dsy, real dsz, real L, const LJModel<real> &m, real hs_i,
                                           real te_i, real hs_j, real te_j, real &fx, real &fy, real &fz, real &e,
                                           real &w) {
    const real rx = L * (dsx - rint(dsx)), ry = L * (dsy - rint(dsy)), rz = L * (dsz - rint(dsz));
    const real r2 = rx * rx + ry * ry + rz * rz;
    if (MODE == EMDEE_CUTOFF && !(r2 < m.rc2)) return;
    const real inv_r2 = fast_rcp(r2);
    real E, W;
    lj_interaction(r2, inv_r2, m, hs_i, te_i, hs_j, te_j, E, W);
    const real wr2 = W * inv_r2;
    fx += wr2 * rx; fy += wr2 * ry; fz += wr2 * rz;
    e += E; w += W;
}

template <typename real, int MODE>
__global__ __launch_bounds__(TILE_BLOCK) void k_tiles(int n, const real *__restrict__ pos, real L,
                                                      const emdee_lj_atom *__restrict__ atoms, LJModel<real> model,
                                                      int bitmask, real *__restrict__ forces,
                                                      real *__restrict__ energies, real *__restrict__ virials) {
    __shared__ real s_part[TILE_WAVES][5][TILE];
    const int lane = threadIdx.x & (WAVE - 1), wv = threadIdx.x / WAVE;
    const int ntiles = (n + TILE - 1) / TILE;
    const int bI = blockIdx.x;
    const int I = bI * TILE + lane;
    const bool vi = I < n;
    real sxi = 0, syi = 0, szi = 0, hs_i = 0, te_i = 0;
    if (vi) {
        sxi = pos[3 * (size_t)I] / L; syi = pos[3 * (size_t)I + 1] / L; szi = pos[3 * (size_t)I + 2] / L;
        hs_i = (real)atoms[I].half_sigma; te_i = (real)atoms[I].twice_sqrt_eps;
    }
    real fx = 0, fy = 0, fz = 0, e = 0, w = 0;
    for (int bJ = wv; bJ < ntiles; bJ += TILE_WAVES) {
        const int J = bJ * TILE + lane;
        const bool vj = J < n;
        real sxj = 0, syj = 0, szj = 0, hs_j = 0, te_j = 0;
        if (vj) {
            sxj = pos[3 * (size_t)J] / L; syj = pos[3 * (size_t)J + 1] / L; szj = pos[3 * (size_t)J + 2] / L;
            hs_j = (real)atoms[J].half_sigma; te_j = (real)atoms[J].twice_sqrt_eps;
        }
        const int m0 = (bJ == bI) ? 1 : 0;   // diagonal tile: skip the self pair
        for (int m = m0; m < TILE; m++) {
            const int src = (lane + m) & (WAVE - 1);
            const real xj = __shfl(sxj, src), yj = __shfl(syj, src), zj = __shfl(szj, src);
            const real hj = __shfl(hs_j, src), tj = __shfl(te_j, src);
            const int ok = __shfl((int)vj, src);
            if (vi && ok) pair_terms<real, MODE>(sxi - xj, syi - yj, szi - zj, L, model, hs_i, te_i, hj, tj, fx, fy, fz, e, w);
        }
    }
    s_part[wv][0][lane] = fx; s_part[wv][1][lane] = fy; s_part[wv][2][lane] = fz;
    s_part[wv][3][lane] = e; s_part[wv][4][lane] = w;
    __syncthreads();
    if (wv == 0 && vi) {
        real t[5];
        for (int q = 0; q < 5; q++) {
            t[q] = 0;
            for (int k = 0; k < TILE_WAVES; k++) t[q] += s_part[k][q][lane];   // fixed order: reproducible
        }
        if (bitmask & EMDEE_FORCES) {
            forces[3 * (size_t)I] = t[0]; forces[3 * (size_t)I + 1] = t[1]; forces[3 * (size_t)I + 2] = t[2];
        }
        if (bitmask & EMDEE_ENERGIES) energies[I] = (real)0.5 * t[3];
        if (bitmask & EMDEE_VIRIALS) virials[I] = (real)0.5 * t[4];
    }
}

template <typename real, int MODE>
__global__ void k_naive(int n, const real *__restrict__ pos, real L, const emdee_lj_atom *__restrict__ atoms,
                        LJModel<real> model, real *__restrict__ forces, real *__restrict__ energies,
                        real *__restrict__ virials) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const real sxi = pos[3 * (size_t)i] / L, syi = pos[3 * (size_t)i + 1] / L, szi = pos[3 * (size_t)i + 2] / L;
    const real hs_i = (real)atoms[i].half_sigma, te_i = (real)atoms[i].twice_sqrt_eps;
    real fx = 0, fy = 0, fz = 0, e = 0, w = 0;
    for (int j = 0; j < n; j++) {
        if (j == i) continue;
        const real sxj = pos[3 * (size_t)j] / L, syj = pos[3 * (size_t)j + 1] / L, szj = pos[3 * (size_t)j + 2] / L;
        pair_terms<real, MODE>(sxi - sxj, syi - syj, szi - szj, L, model, hs_i, te_i, (real)atoms[j].half_sigma,
                               (real)atoms[j].twice_sqrt_eps, fx, fy, fz, e, w);
    }
    forces[3 * (size_t)i] = fx; forces[3 * (size_t)i + 1] = fy; forces[3 * (size_t)i + 2] = fz;
    energies[i] = (real)0.5 * e;
    virials[i] = (real)0.5 * w;
}

template <typename real>
__global__ void k_interaction(int n, const real *__restrict__ r2, LJModel<real> model, emdee_lj_atom ai, emdee_lj_atom aj,
                              int mode, real *__restrict__ E, real *__restrict__ W) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    const real x = r2[k];
    real e = 0, w = 0;
    if (!(mode == EMDEE_CUTOFF && !(x < model.rc2)))
        lj_interaction(x, fast_rcp(x), model, (real)ai.half_sigma, (real)ai.twice_sqrt_eps, (real)aj.half_sigma,
                       (real)aj.twice_sqrt_eps, e, w);
    E[k] = e;
    W[k] = w;
}

}  // namespace emdee
