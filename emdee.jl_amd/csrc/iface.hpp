// iface.hpp -- precision-erased interfaces between the C ABI (capi.hip) and the two template
// instantiations (impl_f32.hip, impl_f64.hip), so the translation units compile in parallel.
#pragma once

#include "common.hpp"

namespace emdee {

struct ICells {
    virtual ~ICells() {}
    virtual void update(const void *positions) = 0;
    virtual int M() const = 0;
    virtual void arrays(const int32_t **index, const int32_t **population, const int32_t **start,
                        const int32_t **order) const = 0;
};

struct INbr {
    virtual ~INbr() {}
    virtual void compute(void *forces, void *energies, void *virials, const void *positions, double L,
                         const emdee_lj_model &model, const emdee_lj_atom *atoms, int bitmask) = 0;
    virtual void stats(int64_t *builds, int64_t *listed, int32_t *max_count, int32_t *capacity) = 0;
    virtual void count_pairs(int64_t *pairs) = 0;
    virtual void export_list(int32_t *counts, int32_t *neighbors, int32_t capacity) = 0;
    virtual void set_pairs(const int32_t *pairs, int32_t n_pairs, bool one_four, double lj14scale) = 0;
};

struct IMd {
    virtual ~IMd() {}
    virtual void set_state(int n_owned, int n_ghost, const void *pos, const void *vel, const emdee_lj_atom *atoms,
                           const void *inv_mass) = 0;
    virtual void get_state(void *pos, void *vel, void *frc, void *en, void *vir) = 0;
    virtual void step(int nsteps, double dt, int rebuild_every) = 0;
    virtual void kick_drift(double dt, double kick) = 0;
    virtual void forces(int bitmask, int phase) = 0;
    virtual void kick(double dt) = 0;
    virtual bool fused_step(double dt, double kick, int phase) = 0;
    virtual bool needs_rebuild() = 0;
    virtual void rebuild() = 0;
    virtual void pack_positions(const int32_t *ids, const int32_t *codes, int n, const double *shifts, int n_shifts,
                                void *buf) = 0;
    virtual void unpack_ghosts(const void *buf, int first, int n) = 0;
    virtual void energies(double out[3]) = 0;
    virtual void stats(int64_t *builds, int64_t *listed, int32_t *max_count, int32_t *capacity) = 0;
    virtual void count_pairs(int64_t *pairs) = 0;
    virtual void export_list(int32_t *counts, int32_t *neighbors, int32_t capacity) = 0;
    virtual void profile(bool enable) = 0;
    virtual void kernel_time(int kernel, double *total_ms, int64_t *launches) = 0;
    virtual void set_langevin(double gamma, double temperature, uint64_t seed, uint64_t first_step) = 0;
    virtual void set_langevin_ids(const int64_t *ids) = 0;
    virtual void langevin_normals(uint64_t seed, uint64_t step, const int64_t *ids, int n, double *out) = 0;
    virtual void set_pairs(const int32_t *pairs, int32_t n_pairs, bool one_four, double lj14scale) = 0;
};

// spatial domain decomposition (emdee_dd_*): the domains of the decomposition that live in this process
struct IDd {
    virtual ~IDd() {}
    virtual void set_atoms(int local, int n, const void *pos, const void *vel, const emdee_lj_atom *atoms, const int64_t *gids) = 0;
    virtual void load() = 0;
    virtual void step(int nsteps, double dt, int rebuild_every) = 0;
    virtual void energies(double out[3]) = 0;
    virtual int64_t n_atoms_global() = 0;
    virtual int n_owned(int local) = 0;
    virtual int n_ghost(int local) = 0;
    virtual IMd *engine(int local) = 0;
    virtual void get_state(int local, int64_t *gids, void *pos, void *vel, void *frc) = 0;
    virtual void set_langevin(double gamma, double temperature, uint64_t seed, uint64_t first_step) = 0;
    virtual void stats(int64_t out[4]) = 0;
    virtual void rebuild_stats(int64_t out[4]) = 0;
    virtual void phase_times(double out[8]) = 0;
    virtual void set_overlap(bool on) = 0;
};

void dd_rccl_selftest(emdee_ctx *ctx, int n_bytes);
// host-only description of one domain's geometry (dd.hpp: DdGeom), for emdee_dd_describe
void dd_describe(const double len[3], const int32_t grid[3], double halo, int rank, int32_t *ndirs, int32_t *dirs,
                 int32_t *dir_rank, double *dir_shift, int32_t *npeers, int32_t *peers, double *local_lo, double *local_len,
                 int32_t *periodic);

template <typename real>
struct Factory {
    static IDd *dd(emdee_ctx *ctx, const double len[3], const int32_t grid[3], int rank_first, int n_local,
                   const void *unique_id, const emdee_lj_model &model, double skin);
    static ICells *cells(emdee_ctx *ctx, int N, double L, double cutoff, int ndiv);
    static INbr *nbr(emdee_ctx *ctx, int N, double skin);
    static IMd *md(emdee_ctx *ctx, const double lo[3], const double len[3], const int32_t per[3],
                   const emdee_lj_model &model, double skin);
    static void tiles(emdee_ctx *ctx, void *f, void *e, void *w, const void *pos, double L, int N,
                      const emdee_lj_model &model, const emdee_lj_atom *atoms, int bitmask, int mode);
    static void naive(emdee_ctx *ctx, void *f, void *e, void *w, const void *pos, double L, int N,
                      const emdee_lj_model &model, const emdee_lj_atom *atoms, int mode);
    static void interaction(emdee_ctx *ctx, int n, const void *r2, const emdee_lj_model &model, emdee_lj_atom ai,
                            emdee_lj_atom aj, int mode, void *E, void *W);
};

}  // namespace emdee
