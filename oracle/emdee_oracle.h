/*
 * emdee_oracle.h -- CPU restatement of EmDee.jl's nonbonded pair-force path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it,
 * and there only as the checker / the timed CPU baseline.
 *
 * PARITY UNPINNED (in the golden-vector sense): the reference (Julia + CUDA.jl)
 * cannot run in this image and its tests hold no stored outputs -- only the
 * input fixture test/data/lj_sample.xyz, the parameters L=10, rc=3, rs=2.5,
 * eps=sigma=1 and a 1e-4 agreement bound between its two implementations
 * (test/runtests.jl:19-42,58).  This restatement is therefore pinned by
 *   (1) closed-form Lennard-Jones / quintic-switch known answers evaluated in
 *       exact rational arithmetic (tests/golden/make_golden.py),
 *   (2) an independent numpy fp64 restatement of the same reference lines on the
 *       reference's own fixture (tests/golden/lj_sample_*.npz),
 *   (3) the survey-time values recorded in SURVEY.md 8(c).
 *
 * Every function cites the reference file:line it follows (paths under the
 * upstream repository root).
 */
#ifndef EMDEE_ORACLE_H
#define EMDEE_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* output-selection bitmask, src/nonbonded.jl:12-14 */
#define ORC_FORCES   1
#define ORC_ENERGIES 2
#define ORC_VIRIALS  4

/* Q1 of SURVEY.md 2.4: the reference formula gives g = 1 (full LJ) beyond rc.
 * LITERAL keeps that; CUTOFF drops pairs with r2 >= rc2 (what any cell-list
 * path can compute). Inside r2 < rc2 both are identical. */
#define ORC_LITERAL 0
#define ORC_CUTOFF  1

/* src/lennard_jones.jl:6-11 -- {rc^2, rs^2, 1/(rc^2-rs^2)} */
typedef struct { float  rc2, rs2, inv_delta2; } orc_model32;
typedef struct { double rc2, rs2, inv_delta2; } orc_model64;
/* src/lennard_jones.jl:13-18 -- LJAtom{half_sigma, twice_sqrt_eps}, always Float32 */
typedef struct { float half_sigma, twice_sqrt_eps; } orc_atom;

void orc_model_f32(double cutoff, double sw, orc_model32 *m);
void orc_model_f64(double cutoff, double sw, orc_model64 *m);
void orc_lj_atom(double eps, double sigma, orc_atom *a);

/* src/lennard_jones.jl:25-42 -- returns (E*g, W*g + E*(-r g')) */
void orc_interaction_f32(float r2, const orc_model32 *m, orc_atom ai, orc_atom aj,
                         int mode, float *E, float *W);
void orc_interaction_f64(double r2, const orc_model64 *m, orc_atom ai, orc_atom aj,
                         int mode, double *E, double *W);

/* src/nonbonded.jl:122-155 -- all pairs i<j; pos is 3xN column-major (xyz
 * interleaved); per-atom energies/virials get half of each pair term.
 * _f32 keeps the reference's mixed accumulation (Q6); _f64 is all double. */
void orc_naive_f32(int32_t N, const float *pos, float L, const orc_model32 *m,
                   const orc_atom *atoms, int mode,
                   float *forces, float *energies, float *virials);
void orc_naive_f64(int32_t N, const double *pos, double L, const orc_model64 *m,
                   const orc_atom *atoms, int mode,
                   double *forces, double *energies, double *virials);

/* src/cells.jl:36,180-181 -- M = floor(ndiv*L/cutoff); 1-based cell id
 * 1 + vx + M*vy + M^2*vz with v = floor(M*(s - floor(s))), s = r/L.
 * index[N], population[M^3] (population may be NULL). Returns M. */
int32_t orc_cells_per_dimension(double L, double cutoff, int32_t ndiv);
int32_t orc_cells_f64(int32_t N, const double *pos, double L, double cutoff, int32_t ndiv,
                      int32_t *index, int32_t *population);
int32_t orc_cells_f32(int32_t N, const float *pos, double L, double cutoff, int32_t ndiv,
                      int32_t *index, int32_t *population);

/* Full neighbour list (both i->j and j->i) of all minimum-image pairs with
 * r2 < rlist^2, by a CPU cell list (intent of src/cells.jl:224-297, made
 * complete). Two-call pattern: offsets[N+1] is always filled; nbrs may be NULL
 * on the first call. Neighbours of each atom are sorted ascending. Returns the
 * total number of entries, or -1 if rlist > L/2. */
int64_t orc_neighbor_list_f64(int32_t N, const double *pos, double L, double rlist,
                              int64_t *offsets, int32_t *nbrs);

/* Same arithmetic as orc_naive_f64 in CUTOFF mode, O(N) via a cell list;
 * OpenMP over atoms (nthreads <= 0: all cores). Owner-computes full list. */
void orc_nonbonded_cells_f64(int32_t N, const double *pos, double L, const orc_model64 *m,
                             const orc_atom *atoms, int nthreads,
                             double *forces, double *energies, double *virials);
void orc_nonbonded_cells_f32(int32_t N, const float *pos, float L, const orc_model32 *m,
                             const orc_atom *atoms, int nthreads,
                             float *forces, float *energies, float *virials);

/* Build-defined velocity-Verlet (absent from the reference):
 *   v += (dt/2) f/m ; x += dt v ; f = F(x) ; v += (dt/2) f/m
 * x,v are 3xN in/out; inv_mass may be NULL (m = 1). use_cells != 0 selects the
 * O(N) force path (CUTOFF), otherwise all-pairs CUTOFF. epot/ekin/virial
 * (each nsteps+1 long, may be NULL) receive totals at steps 0..nsteps.
 * forces_out (3N, may be NULL) receives the final forces. */
void orc_verlet_f64(int32_t N, double *x, double *v, double L, const orc_model64 *m,
                    const orc_atom *atoms, const double *inv_mass, double dt,
                    int32_t nsteps, int use_cells, int nthreads,
                    double *epot, double *ekin, double *virial, double *forces_out);

/* Langevin thermostat, build-defined.  Three N(0,1) numbers per (seed, step, atom id) from
 * splitmix64-finalised counters and Box-Muller; the thermostatted step is
 *   v += (dt/2) f/m ; v = c1 v + c2 sqrt(T/m) xi ; x += dt v ; f = F(x) ; v += (dt/2) f/m,
 * c1 = exp(-gamma dt), c2 = sqrt(1 - c1^2).  ids may be NULL (id = index). */
void orc_langevin_normals(uint64_t seed, uint64_t step, uint64_t id, double out[3]);
void orc_verlet_langevin_f64(int32_t N, double *x, double *v, double L, const orc_model64 *m,
                             const orc_atom *atoms, const double *inv_mass, double dt, int32_t nsteps,
                             int use_cells, int nthreads, double gamma, double temperature, uint64_t seed,
                             uint64_t step0, const int64_t *ids, double *epot, double *ekin, double *virial,
                             double *forces_out);

int orc_max_threads(void);

#ifdef __cplusplus
}
#endif
#endif
