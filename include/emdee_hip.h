/*
 * emdee_hip.h -- C ABI of libemdee_hip.so: the MI355X (gfx950) implementation of
 * EmDee.jl's nonbonded pair-force hot path (cell-list neighbour build, switched
 * Lennard-Jones force / per-atom energy / per-atom virial, velocity-Verlet).
 *
 * This is the drop-in boundary.  EmDee's Julia operator layer (src/lennard_jones.jl,
 * src/nonbonded.jl, src/cells.jl of the reference) binds these symbols with `ccall`
 * in place of its CUDA.jl kernels; INTEGRATION.md shows that binding, and
 * emdee.jl_amd/ mirrors the same operator API in Python over ctypes.
 *
 * Conventions
 *   - extern "C", plain pointers and sizes; no C++/torch types cross the boundary.
 *   - Every call returns int32 status: 0 = EMDEE_OK, negative = error; the message of
 *     the last error on the calling thread is emdee_last_error().  No exception crosses.
 *   - Handles are opaque, owned by the library, freed by the matching *_destroy.
 *   - "dev" pointers are device (HBM) pointers owned by the caller (hipMalloc,
 *     emdee_malloc or a torch tensor's data_ptr) on the context's device.
 *   - Work is enqueued on the context's HIP stream; emdee_sync() and the calls that
 *     return host values are the only blocking calls.  One host thread per context,
 *     as in the reference (single-threaded host, async launches, src/nonbonded.jl:115-119).
 *   - Streams.  A decomposed domain (emdee_dd_*) and the integrator it lends out (emdee_dd_engine)
 *     run on streams the library owns.  The calls that copy an engine's state into caller arrays --
 *     emdee_dd_get_state, emdee_md_get_state, emdee_md_nbr_list -- order themselves against the
 *     caller's context stream inside the library: the engine's stream waits for what the caller's
 *     stream had queued when the call was made (so a block the caller's allocator has just recycled is
 *     not written early), and the caller's stream waits for the copies (so work queued on it after the
 *     call sees them).  No device-wide synchronise, nothing for a binding to add.  Arrays handed IN
 *     (emdee_dd_set_atoms, emdee_md_create, ...) are read on the context's stream, in order.
 *   - Arrays follow the reference: positions/forces/velocities are 3xN column-major
 *     (xyz interleaved, src/nonbonded.jl:52-61), energies/virials length N.
 *   - precision = bytes per real of the caller's arrays: EMDEE_F32 (the reference's
 *     Float32) or EMDEE_F64 (north-star fp64).  Pair math runs in that type.
 *   - Pair semantics are the reference formula src/lennard_jones.jl:25-42.  The O(N)
 *     path drops pairs with r^2 >= rc^2 (EMDEE_CUTOFF); the all-pairs entry points can
 *     also reproduce the reference's literal behaviour, where the switch clamp gives
 *     g = 1 (full LJ) beyond rc (EMDEE_LITERAL).
 */
#ifndef EMDEE_HIP_H
#define EMDEE_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define EMDEE_VERSION 100            /* 0.1.0, tracks the reference's Project.toml:4 */

/* status codes */
#define EMDEE_OK                 0
#define EMDEE_ERR_INVALID       -1   /* bad argument */
#define EMDEE_ERR_HIP           -2   /* HIP runtime error (message has hipGetErrorString) */
#define EMDEE_ERR_NO_DEVICE     -3   /* no usable gfx950 device */
#define EMDEE_ERR_ALLOC         -4
#define EMDEE_ERR_OVERFLOW      -5   /* neighbour capacity could not be grown */
#define EMDEE_ERR_STATE         -6   /* call out of order (e.g. step before set_state) */

/* output-selection bitmask -- src/nonbonded.jl:12-14 */
#define EMDEE_FORCES    1
#define EMDEE_ENERGIES  2
#define EMDEE_VIRIALS   4

/* precision = sizeof(real) */
#define EMDEE_F32 4
#define EMDEE_F64 8

/* pair semantics beyond the cutoff (all-pairs entry points only) */
#define EMDEE_LITERAL 0
#define EMDEE_CUTOFF  1

/* LennardJonesModel -- src/lennard_jones.jl:6-11: {rc^2, rs^2, 1/(rc^2 - rs^2)}.
 * Passed by value as doubles; the EMDEE_F32 path rounds each field to float first,
 * which is exactly the reference's Float32 struct. */
typedef struct emdee_lj_model { double rc2, rs2, inv_delta2; } emdee_lj_model;

/* LJAtom -- src/lennard_jones.jl:15-18, identical layout (2 x Float32, 8 bytes).
 * sigma_ij = half_sigma_i + half_sigma_j, 4 eps_ij = twice_sqrt_eps_i * twice_sqrt_eps_j
 * (Lorentz-Berthelot), src/lennard_jones.jl:29-30. */
typedef struct emdee_lj_atom { float half_sigma, twice_sqrt_eps; } emdee_lj_atom;

typedef struct emdee_ctx   emdee_ctx;     /* device + stream                      (replaces CUDA.jl context handling) */
typedef struct emdee_cells emdee_cells;   /* Cells                                (src/cells.jl:6-20) */
typedef struct emdee_nbr   emdee_nbr;     /* neighbour handle carried by `tiles`  (src/nonbonded.jl:18-26) */
typedef struct emdee_md    emdee_md;      /* velocity-Verlet state                (absent from the reference) */
typedef struct emdee_dd    emdee_dd;      /* spatial domain decomposition         (absent from the reference) */

/* ---------------------------------------------------------------- context */
const char *emdee_last_error(void);
int32_t emdee_version(void);
int32_t emdee_device_count(int32_t *count);
/* stream: a hipStream_t to enqueue on (e.g. torch's current stream), or NULL for the
 * device's default stream. */
int32_t emdee_ctx_create(int32_t device_id, void *stream, emdee_ctx **out);
int32_t emdee_ctx_destroy(emdee_ctx *ctx);
int32_t emdee_sync(emdee_ctx *ctx);
/* arch name ("gfx950..."), CU count and HBM bytes of the context's device */
int32_t emdee_device_info(emdee_ctx *ctx, char *arch, size_t arch_len, int32_t *cu_count, int64_t *hbm_bytes);

/* ---------------------------------------------------------------- memory
 * Replaces CuArray / CUDA.cu / CUDA.zeros / Array(dev) / unsafe_copyto!
 * (src/nonbonded.jl:25,123,151-153; test/runtests.jl:22-35). */
int32_t emdee_malloc(emdee_ctx *ctx, size_t nbytes, void **dev);
int32_t emdee_free(emdee_ctx *ctx, void *dev);
int32_t emdee_memcpy_h2d(emdee_ctx *ctx, void *dev, const void *host, size_t nbytes);
int32_t emdee_memcpy_d2h(emdee_ctx *ctx, void *host, const void *dev, size_t nbytes);   /* blocking */
int32_t emdee_memcpy_d2d(emdee_ctx *ctx, void *dst, const void *src, size_t nbytes);
int32_t emdee_memset(emdee_ctx *ctx, void *dev, int32_t byte, size_t nbytes);

/* ---------------------------------------------------------------- pair function
 * interaction(r2, model, atom_i, atom_j) -- src/lennard_jones.jl:25-42, evaluated by the
 * DEVICE pair function on n values of r2 (dev arrays of `precision` reals): E[k], W[k]. */
int32_t emdee_interaction(emdee_ctx *ctx, int32_t n, const void *r2_dev, emdee_lj_model model,
                          emdee_lj_atom atom_i, emdee_lj_atom atom_j, int32_t mode,
                          void *E_dev, void *W_dev, int32_t precision);

/* ---------------------------------------------------------------- Cells
 * Cells(r, L, cutoff; ndiv=2) -- src/cells.jl:176-194: M = floor(ndiv L / cutoff) cells per
 * dimension, 1-based cell id 1 + vx + M vy + M^2 vz, v = floor(M (s - floor s)), s = r / L.
 * emdee_cells_update = update_cells!(cells, r, L) (src/cells.jl:196-222); it re-bins all
 * atoms in O(N) (counting sort) instead of the reference's incremental linked-list edit. */
int32_t emdee_cells_create(emdee_ctx *ctx, int32_t N, double L, double cutoff, int32_t ndiv,
                           int32_t precision, emdee_cells **out);
int32_t emdee_cells_update(emdee_cells *cells, const void *positions_dev);
int32_t emdee_cells_destroy(emdee_cells *cells);
int32_t emdee_cells_M(const emdee_cells *cells, int32_t *M);
/* dev pointers valid until the next update/destroy: index[N] (1-based cell of each atom),
 * population[M^3], start[M^3+1] (0-based offsets into order), order[N] (atom ids by cell,
 * ascending within a cell -- the array form of the reference's head/next lists). */
int32_t emdee_cells_arrays(const emdee_cells *cells, const int32_t **index_dev,
                           const int32_t **population_dev, const int32_t **start_dev,
                           const int32_t **order_dev);

/* ---------------------------------------------------------------- neighbour handle
 * nonbonded_computation_tiles(N) (src/nonbonded.jl:18-26) returns what compute_nonbonded!
 * iterates over.  Here that object is a neighbour-list workspace for N atoms: cell binning,
 * cell-ordered copies and a full (owner-computes) list with a skin, rebuilt inside
 * emdee_compute_nonbonded when any atom has moved more than skin/2 since the last build. */
int32_t emdee_nbr_create(emdee_ctx *ctx, int32_t N, double skin, int32_t precision, emdee_nbr **out);
int32_t emdee_nbr_destroy(emdee_nbr *nbr);
/* builds = list builds so far; listed = entries in the current list; max_count = longest
 * row; capacity = row stride.  Blocking. */
int32_t emdee_nbr_stats(emdee_nbr *nbr, int64_t *builds, int64_t *listed, int32_t *max_count,
                        int32_t *capacity);
/* number of pairs with r^2 < rc^2 in the current list (each pair counted once). Blocking. */
int32_t emdee_nbr_count_pairs(emdee_nbr *nbr, int64_t *pairs_in_cutoff);

/* Verification accessor: the current list as CALLER ids.  counts_dev[i] = entries of atom i's row (skin entries
 * included), neighbors_dev[i * capacity + k] = its k-th neighbour, k < min(counts, capacity); capacity >= the
 * `capacity` of emdee_nbr_stats returns every row whole.  Owned atoms only; ghosts appear as neighbours.
 * (The list itself stores 16-bit tile-local slots; this decodes them through the brick tables.) */
int32_t emdee_nbr_list(emdee_nbr *nbr, int32_t *counts_dev, int32_t *neighbors_dev, int32_t capacity);
int32_t emdee_md_nbr_list(emdee_md *md, int32_t *counts_dev, int32_t *neighbors_dev, int32_t capacity);

/* Exclusions and 1-4 pairs (build-defined; SURVEY.md 8(f) item 2 "hooks": the reference parses lj14scale from its force-field
 * file, src/modelling.jl:197-200, and nothing consumes it -- its hot path sums every pair, src/nonbonded.jl:129-150).
 * pairs_dev: n_pairs pairs {i, j} of atom indices (caller order, 2 n_pairs int32, device); copied.  A pair named by
 * emdee_*_set_exclusions contributes nothing; a pair named by emdee_*_set_pairs14 contributes lj14scale times its pair terms
 * (forces, energy and virial halves).  Both are struck from the neighbour rows right after every list build -- the pair loop
 * carries no mask -- and the 1-4 pairs are evaluated by a kernel of their own behind every force pass (an integrator with
 * 1-4 pairs steps with the split kernels: force pass, 1-4 terms, kick + drift).  Each call replaces its table; n_pairs = 0
 * clears it.  Undivided boxes (emdee_nbr, emdee_md without ghosts); emdee_md: after emdee_md_set_state.  Two-species boxes with
 * exclusions keep the general-species kernels. */
int32_t emdee_nbr_set_exclusions(emdee_nbr *nbr, const int32_t *pairs_dev, int32_t n_pairs);
int32_t emdee_nbr_set_pairs14(emdee_nbr *nbr, const int32_t *pairs_dev, int32_t n_pairs, double lj14scale);
int32_t emdee_md_set_exclusions(emdee_md *md, const int32_t *pairs_dev, int32_t n_pairs);
int32_t emdee_md_set_pairs14(emdee_md *md, const int32_t *pairs_dev, int32_t n_pairs, double lj14scale);

/* compute_nonbonded!(forces, energies, virials, positions, L, tiles, model, atoms, Val(bitmask))
 * -- src/nonbonded.jl:109-120 -- O(N) neighbour-list path, EMDEE_CUTOFF semantics.
 * Outputs not selected by bitmask may be NULL and are left untouched; selected outputs are
 * overwritten (the reference zero-fills then accumulates, src/nonbonded.jl:112-114). */
int32_t emdee_compute_nonbonded(emdee_ctx *ctx, void *forces_dev, void *energies_dev, void *virials_dev,
                                const void *positions_dev, double L, emdee_nbr *nbr,
                                emdee_lj_model model, const emdee_lj_atom *atoms_dev,
                                int32_t bitmask, int32_t precision);

/* compute_tile! semantics (src/nonbonded.jl:44-107) for any N: all-pairs 64x64 tiles,
 * one wavefront per tile pair, lane rotation through DPP/bpermute instead of shfl_sync.
 * mode = EMDEE_LITERAL reproduces the reference operator exactly (Q1). */
int32_t emdee_compute_nonbonded_tiles(emdee_ctx *ctx, void *forces_dev, void *energies_dev, void *virials_dev,
                                      const void *positions_dev, double L, int32_t N,
                                      emdee_lj_model model, const emdee_lj_atom *atoms_dev,
                                      int32_t bitmask, int32_t mode, int32_t precision);

/* naively_compute_nonbonded!(forces, energies, virials, positions, L, model, atoms)
 * -- src/nonbonded.jl:122-155 -- the plain double loop, one device thread per atom i over
 * all j != i (no tiles, no list, no lane exchange); always all three outputs. */
int32_t emdee_compute_nonbonded_naive(emdee_ctx *ctx, void *forces_dev, void *energies_dev, void *virials_dev,
                                      const void *positions_dev, double L, int32_t N,
                                      emdee_lj_model model, const emdee_lj_atom *atoms_dev,
                                      int32_t mode, int32_t precision);

/* ---------------------------------------------------------------- velocity-Verlet
 * Build-defined:  v += (dt/2m) f ; x += dt v ; f = F(x) ; v += (dt/2m) f.
 * The state lives on the device in cell order between calls.  The box is orthorhombic
 * [lo, lo+len) per dimension; periodic[d] != 0 applies the minimum-image convention along d.
 * Atoms 0..n_owned-1 are integrated; atoms n_owned..n_owned+n_ghost-1 are ghosts (images
 * owned by another domain): they act on owned atoms but receive no force and
 * are moved only by emdee_md_unpack_ghosts.  The single-GPU reference-shaped case is
 * lo = 0, len = L, periodic = {1,1,1}, n_ghost = 0. */
int32_t emdee_md_create(emdee_ctx *ctx, const double lo[3], const double len[3], const int32_t periodic[3],
                        emdee_lj_model model, double skin, int32_t precision, emdee_md **out);
int32_t emdee_md_destroy(emdee_md *md);
/* (Re)load the state, in caller order.  velocities_dev has 3 n_owned reals; inv_mass_dev
 * (n_owned reals) may be NULL for m = 1 (LJAtom has no mass, src/lennard_jones.jl:15-18).
 * Bins, sorts, builds the neighbour list and evaluates the forces. */
int32_t emdee_md_set_state(emdee_md *md, int32_t n_owned, int32_t n_ghost, const void *positions_dev,
                           const void *velocities_dev, const emdee_lj_atom *atoms_dev,
                           const void *inv_mass_dev);
/* Copy the state back in caller order; any pointer may be NULL. positions: n_owned+n_ghost
 * atoms; velocities/forces/energies/virials: n_owned atoms. */
int32_t emdee_md_get_state(emdee_md *md, void *positions_dev, void *velocities_dev, void *forces_dev,
                           void *energies_dev, void *virials_dev);
/* nsteps whole steps (n_ghost must be 0: a decomposed run drives the split calls below).
 * rebuild_every > 0: fixed cadence; 0: rebuild when max displacement > skin/2.
 * Every inner step is one kernel (force + kick + drift).  With the displacement trigger the steps are
 * queued a few at a time and read back once per batch; a step queued behind one that asked for a
 * rebuild checks a device word first and does nothing, so the states produced are exactly those of
 * stepping one at a time, and a given sequence of calls is bitwise reproducible from run to run
 * (deterministic cell order, owner-computes sums, no floating-point atomics). */
int32_t emdee_md_step(emdee_md *md, int32_t nsteps, double dt, int32_t rebuild_every);
/* split step for domain-decomposed runs:  kick_drift -> [halo exchange] -> forces -> kick */
/* v += kick (dt/m) f ; x += dt v (owned).  kick = 0.5: the opening half kick; kick = 1.0 also
 * carries the closing half kick of the previous step (same f), saving one pass over v and f. */
int32_t emdee_md_kick_drift(emdee_md *md, double dt, double kick);
/* f (and e, w) of owned atoms.  phase 0: all atoms; phase 1: only bricks whose LDS tile holds no
 * ghost cell (can run while the halo exchange is in flight); phase 2: the remaining bricks. */
int32_t emdee_md_forces(emdee_md *md, int32_t bitmask, int32_t phase);
int32_t emdee_md_kick(emdee_md *md, double dt);              /* v += (dt/2m) f */
/* One inner step as a single kernel: f = F(x), v += kick (dt/m) f, x += dt v, with the new positions
 * written to the second position buffer.  phase as in emdee_md_forces; the buffers are swapped after
 * phase 0 or phase 2, so a decomposed run calls phase 1 (interior bricks, while the halo exchange of
 * the CURRENT positions is in flight), unpacks the ghosts, then phase 2.  Sets *fused = 0 and does
 * nothing if the LDS-tiled kernels are not in use for this box (caller falls back to the split step). */
int32_t emdee_md_fused_step(emdee_md *md, double dt, double kick, int32_t phase, int32_t *fused);
/* 1 if some owned atom moved more than skin/2 since the last build. Blocking. */
int32_t emdee_md_needs_rebuild(emdee_md *md, int32_t *flag);
/* re-bin, re-sort and rebuild the neighbour list from the current positions */
int32_t emdee_md_rebuild(emdee_md *md);
/* halo: gather positions of the listed atoms (caller-order ids, dev int32[n]) into buf (3 n
 * reals), each plus the periodic-image shift shifts[3 codes[k] .. +2] (codes_dev: dev int32[n],
 * NULL = every atom uses shifts[0..2]; n_shifts <= 27 rows of 3 doubles on the HOST); and scatter
 * received positions into ghost slots first..first+n-1 (ghost-relative). */
int32_t emdee_md_pack_positions(emdee_md *md, const int32_t *ids_dev, const int32_t *codes_dev, int32_t n,
                                const double *shifts, int32_t n_shifts, void *buf_dev);
int32_t emdee_md_unpack_ghosts(emdee_md *md, const void *buf_dev, int32_t first, int32_t n);
/* totals over owned atoms: out[0] = potential energy (sum of per-atom halves), out[1] =
 * kinetic energy, out[2] = virial sum.  Evaluates energies/virials if needed. Blocking. */
int32_t emdee_md_energies(emdee_md *md, double out[3]);
int32_t emdee_md_nbr_stats(emdee_md *md, int64_t *builds, int64_t *listed, int32_t *max_count,
                           int32_t *capacity);
int32_t emdee_md_count_pairs(emdee_md *md, int64_t *pairs_in_cutoff);
/* Per-kernel device time from HIP events recorded on the context's stream while
 * profiling is on.  kernel: 0 = lj_force_nbr (plain force launches), 1 = verlet_kick_drift, 2 = rebuild
 * (bin + sort + nbr_build), 3 = verlet_kick, 4 = lj_force_nbr with the velocity-Verlet update fused in
 * (emdee_md_step's inner steps, emdee_md_fused_step).  Blocking. */
int32_t emdee_md_profile(emdee_md *md, int32_t enable);
int32_t emdee_md_kernel_time(emdee_md *md, int32_t kernel, double *total_ms, int64_t *launches);

/* Langevin thermostat (build-defined like the integrator: the reference has neither).
 * gamma > 0 switches it on for every later step of this integrator, gamma <= 0 off.  Each step becomes
 *   v += (dt/2) f/m ;  v = c1 v + c2 sqrt(T/m) xi ;  x += dt v ;  f = F(x) ;  v += (dt/2) f/m
 * with c1 = exp(-gamma dt), c2 = sqrt(1 - c1^2) and xi three N(0,1) numbers that are a pure function of
 * (seed, step number, atom id): splitmix64-finalised counters + Box-Muller, so a run is reproducible whatever
 * the sort order or the decomposition.  Steps are numbered from first_step (pass the number of steps already
 * done when resuming).  The O step rides in the kick/drift pass (the fused step kernel or verlet_kick_drift);
 * the noise itself comes from one extra streaming kernel per step. */
int32_t emdee_md_set_langevin(emdee_md *md, double gamma, double temperature, uint64_t seed, uint64_t first_step);
/* Atom ids for the noise counters: device array in caller order, one int64 per owned atom (decomposed runs pass
 * global ids); NULL = the caller index.  Not copied: must stay valid, and is forgotten by emdee_md_set_state. */
int32_t emdee_md_set_langevin_ids(emdee_md *md, const int64_t *ids_dev);
/* The generator on its own, for tests: out_dev[3 i .. 3 i + 2] = xi(seed, step, ids_dev[i]). */
int32_t emdee_md_langevin_normals(emdee_md *md, uint64_t seed, uint64_t step, const int64_t *ids_dev, int32_t n,
                                  double *out_dev);

/* ---------------------------------------------------------------- domain decomposition (multi-GPU)
 * Build-defined (the reference is single-GPU; SURVEY.md 8(b) table rows emdee_dd_create / emdee_dd_step, 8(e)).
 * The periodic box [0, len_d) is cut into grid[0] x grid[1] x grid[2] bricks (at most 3 per dimension), domain
 * rank = cx + grid[0] (cy + grid[1] cz).  One emdee_dd drives the domains rank_first .. rank_first + n_local - 1:
 *   - one process per GPU: n_local = 1 and unique_id = the 128 bytes of emdee_dd_unique_id() as generated by ONE
 *     process and distributed by the caller (Julia Distributed / MPI / torch.distributed); the halo messages then
 *     travel over RCCL (ncclSend/ncclRecv over xGMI), resolved from librccl.so.1 at run time;
 *   - n_local = the whole grid, unique_id = NULL: every domain in this process on the context's device, messages
 *     as device-to-device copies (validation of a decomposition on a one-GPU box).
 * A step is, per domain, pack -> halo exchange on a communication stream || force + kick + drift of the interior
 * bricks -> unpack -> the boundary bricks; the steps of a call are queued a few at a time with device-side guard
 * words (the rebuild request rides on the halo messages), one host read-back per batch.  Rebuild = migration of
 * the atoms that left their brick + new ghost lists + re-sort + neighbour list.  Trajectories are those of the
 * undivided box to rounding. */
int32_t emdee_dd_unique_id(uint8_t out[128]);
/* Diagnostic: resolve librccl, build a ONE-rank communicator on the context's device and push n_bytes through
 * ncclSend/ncclRecv to itself (one group, a stream of its own) and 3 doubles through ncclAllReduce; fails unless the
 * bytes arrive unchanged.  Exercises the run-time binding (symbols, enum values, ncclUniqueId by value) on a one-GPU box. */
int32_t emdee_dd_rccl_selftest(emdee_ctx *ctx, int32_t n_bytes);
/* Host-only (no device needed): the geometry domain `rank` of the grid works with -- its neighbour directions in the
 * library's fixed order (x fastest; only cut dimensions move), the rank behind each direction and the periodic shift a
 * ghost sent that way carries, the distinct peer ranks (ascending), and the local box handed to the integrator (brick
 * plus a halo of width `halo` along cut dimensions, the whole period otherwise).  Arrays sized for 26 directions. */
int32_t emdee_dd_describe(const double len[3], const int32_t grid[3], double halo, int32_t rank, int32_t *ndirs,
                          int32_t dirs[78], int32_t dir_rank[26], double dir_shift[78], int32_t *npeers, int32_t peers[26],
                          double local_lo[3], double local_len[3], int32_t periodic[3]);
int32_t emdee_dd_create(emdee_ctx *ctx, const double len[3], const int32_t grid[3], int32_t rank_first, int32_t n_local,
                        const uint8_t *unique_id, emdee_lj_model model, double skin, int32_t precision, emdee_dd **out);
int32_t emdee_dd_destroy(emdee_dd *dd);
/* Atoms initially held by local domain `local` (any atoms of the box, anywhere: emdee_dd_load hands each to the brick
 * that contains it).  Device arrays in caller order: positions, velocities 3 x n reals; atoms n; gids n global ids
 * (they key the Langevin noise and identify atoms in emdee_dd_get_state).  Copied. */
int32_t emdee_dd_set_atoms(emdee_dd *dd, int32_t local, int32_t n, const void *positions_dev, const void *velocities_dev,
                           const emdee_lj_atom *atoms_dev, const int64_t *gids_dev);
/* Collective over all domains: migrate, select and exchange ghosts, bin/sort/list, forces. */
int32_t emdee_dd_load(emdee_dd *dd);
/* nsteps velocity-Verlet steps of the whole box (collective).  rebuild_every as in emdee_md_step. */
int32_t emdee_dd_step(emdee_dd *dd, int32_t nsteps, double dt, int32_t rebuild_every);
/* global {potential, kinetic, virial} sums (collective, blocking) */
int32_t emdee_dd_energies(emdee_dd *dd, double out[3]);
/* atoms in the whole box / owned and ghost atoms of a local domain */
int32_t emdee_dd_counts(emdee_dd *dd, int32_t local, int64_t *n_global, int32_t *n_owned, int32_t *n_ghost);
/* owned atoms of a local domain: global ids and positions (wrapped into the global box at the last rebuild),
 * velocities, forces; n_owned entries each, any pointer may be NULL.  Blocking. */
int32_t emdee_dd_get_state(emdee_dd *dd, int32_t local, int64_t *gids_dev, void *positions_dev, void *velocities_dev,
                           void *forces_dev);
/* the integrator of a local domain, for the emdee_md_* queries (profile, kernel_time, nbr_stats, count_pairs);
 * borrowed: valid until emdee_dd_destroy, not to be stepped or destroyed by the caller */
int32_t emdee_dd_engine(emdee_dd *dd, int32_t local, emdee_md **out);
int32_t emdee_dd_set_langevin(emdee_dd *dd, double gamma, double temperature, uint64_t seed, uint64_t first_step);
/* out[0] = rebuilds, out[1] = batches of queued steps, out[2] = queued steps cancelled by a rebuild request,
 * out[3] = atoms that changed owner (this process) */
int32_t emdee_dd_stats(emdee_dd *dd, int64_t out[4]);
/* How the rebuilds of this process went: out[0] = rebuilds whose migrant and ghost rows travelled in capacity-padded
 * messages with no count exchange (one read-back), out[1] = of those, the ones redone with exact counts because a
 * capacity was exceeded on some rank, out[2] = migrant rows a message holds per peer, out[3] = ghost rows the messages
 * of local domain 0 hold in all (send side). */
int32_t emdee_dd_rebuild_stats(emdee_dd *dd, int64_t out[4]);
/* Where the host-side time of this process's decomposition goes, cumulative since emdee_dd_create (take differences
 * around a timed region): out[0] = wall-clock ms inside rebuilds (ownership path, exchanges, read-backs, the engines'
 * sort + list), out[1] = rebuilds (the load included), out[2] = wall-clock ms of blocking read-backs of device words,
 * out[3] = read-backs, out[4] = ghost share n_ghost / (n_owned + n_ghost) of local domain 0 as of now, out[5] = the read-backs
 * among out[3] that happened INSIDE rebuilds (a rebuild in the engines' own order has one, with the build's words; the read-back
 * of a batch's request words that asked for it is not among them), out[6] = rebuilds done in the engines' own order, out[7] =
 * engines loaded again for room (a local matter: the state outgrew the slots its last load left).
 * The device-side phases of a step are emdee_md_kernel_time of emdee_dd_engine: 5 = fused step launches over interior
 * bricks (or all bricks, in-order form), 6 = over boundary bricks, 7 = halo (pack -> exchange -> unpack), 2 = sort + list. */
int32_t emdee_dd_phase_times(emdee_dd *dd, double out[8]);
/* How a step meets its halo exchange.  1 (default): interior bricks while the messages travel on a communication stream,
 * boundary bricks on a stream of their own when they have arrived.  0: pack, exchange, unpack and ONE launch over all
 * bricks, in order on the compute stream -- no events, no split launch; cheaper when the messages are short next to the
 * cross-stream bookkeeping (small domains).  Same results either way; collective (all ranks must choose alike only for
 * speed, not for correctness).  Call between emdee_dd_step calls. */
int32_t emdee_dd_set_overlap(emdee_dd *dd, int32_t overlap);

#ifdef __cplusplus
}
#endif
#endif /* EMDEE_HIP_H */
