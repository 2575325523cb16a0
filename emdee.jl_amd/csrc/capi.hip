// capi.hip -- the extern "C" boundary declared in include/emdee_hip.h: context, memory and
// precision dispatch.  No template code here; see impl.hpp.
#include <dlfcn.h>

#include <mutex>
#include <vector>

#include "iface.hpp"

namespace emdee {

static thread_local char g_error[1024] = "";

void set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_error, sizeof(g_error), fmt, ap);
    va_end(ap);
}
const char *get_error() { return g_error; }

static inline bool valid_precision(int32_t p) { return p == EMDEE_F32 || p == EMDEE_F64; }

}  // namespace emdee

using namespace emdee;

struct emdee_cells { emdee_ctx *ctx; ICells *impl; };
struct emdee_nbr   { emdee_ctx *ctx; INbr *impl; };
struct emdee_md    { emdee_ctx *ctx; IMd *impl; };
struct emdee_dd    { emdee_ctx *ctx; IDd *impl; std::vector<emdee_md *> engines; };

#define REQUIRE_PTR(p, what) EMDEE_REQUIRE((p) != nullptr, EMDEE_ERR_INVALID, what " is NULL")
#define REQUIRE_PRECISION(p) EMDEE_REQUIRE(valid_precision(p), EMDEE_ERR_INVALID, "precision must be EMDEE_F32 (4) or EMDEE_F64 (8), got %d", (int)(p))

extern "C" {

const char *emdee_last_error(void) { return get_error(); }
int32_t emdee_version(void) { return EMDEE_VERSION; }

int32_t emdee_device_count(int32_t *count) {
    return guarded([&] {
        REQUIRE_PTR(count, "count");
        int n = 0;
        hipError_t e = hipGetDeviceCount(&n);
        *count = (e == hipSuccess) ? n : 0;
    });
}

int32_t emdee_ctx_create(int32_t device_id, void *stream, emdee_ctx **out) {
    return guarded([&] {
        REQUIRE_PTR(out, "out");
        *out = nullptr;
        int n = 0;
        hipError_t e = hipGetDeviceCount(&n);
        EMDEE_REQUIRE(e == hipSuccess && n > 0, EMDEE_ERR_NO_DEVICE, "no HIP device: %s (this library has no CPU path)",
                      e == hipSuccess ? "device count is 0" : hipGetErrorString(e));
        EMDEE_REQUIRE(device_id >= 0 && device_id < n, EMDEE_ERR_INVALID, "device_id %d out of range [0,%d)", device_id, n);
        EMDEE_HIP_CHECK(hipSetDevice(device_id));
        hipDeviceProp_t prop;
        EMDEE_HIP_CHECK(hipGetDeviceProperties(&prop, device_id));
        EMDEE_REQUIRE(strncmp(prop.gcnArchName, "gfx950", 6) == 0, EMDEE_ERR_NO_DEVICE,
                      "device %d is %s; libemdee_hip is built for gfx950 (MI355X) only", device_id, prop.gcnArchName);
        emdee_ctx *ctx = new emdee_ctx();
        ctx->device = device_id;
        ctx->stream = (hipStream_t)stream;
        ctx->owns_stream = false;
        ctx->cu_count = prop.multiProcessorCount;
        ctx->hbm_bytes = prop.totalGlobalMem;
        snprintf(ctx->arch, sizeof(ctx->arch), "%s", prop.gcnArchName);
        try {
            host_words_alloc(ctx);
        } catch (...) {
            delete ctx;
            throw;
        }
        *out = ctx;
    });
}

int32_t emdee_ctx_destroy(emdee_ctx *ctx) {
    return guarded([&] {
        if (!ctx) return;
        (void)hipSetDevice(ctx->device);
        (void)hipStreamSynchronize(ctx->stream);
        if (ctx->host_flags) (void)hipHostFree(ctx->host_flags);
        delete ctx;
    });
}

int32_t emdee_sync(emdee_ctx *ctx) {
    return guarded([&] {
        REQUIRE_PTR(ctx, "ctx");
        use_device(ctx);
        EMDEE_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    });
}

int32_t emdee_device_info(emdee_ctx *ctx, char *arch, size_t arch_len, int32_t *cu_count, int64_t *hbm_bytes) {
    return guarded([&] {
        REQUIRE_PTR(ctx, "ctx");
        if (arch && arch_len) snprintf(arch, arch_len, "%s", ctx->arch);
        if (cu_count) *cu_count = ctx->cu_count;
        if (hbm_bytes) *hbm_bytes = (int64_t)ctx->hbm_bytes;
    });
}

// ---------------------------------------------------------------- memory
int32_t emdee_malloc(emdee_ctx *ctx, size_t nbytes, void **dev) {
    return guarded([&] {
        REQUIRE_PTR(ctx, "ctx");
        REQUIRE_PTR(dev, "dev");
        use_device(ctx);
        *dev = nullptr;
        if (nbytes == 0) return;
        hipError_t e = hipMalloc(dev, nbytes);
        EMDEE_REQUIRE(e == hipSuccess, EMDEE_ERR_ALLOC, "hipMalloc(%zu) failed: %s", nbytes, hipGetErrorString(e));
    });
}

int32_t emdee_free(emdee_ctx *ctx, void *dev) {
    return guarded([&] {
        REQUIRE_PTR(ctx, "ctx");
        use_device(ctx);
        if (dev) EMDEE_HIP_CHECK(hipFree(dev));
    });
}

int32_t emdee_memcpy_h2d(emdee_ctx *ctx, void *dev, const void *host, size_t nbytes) {
    return guarded([&] {
        REQUIRE_PTR(ctx, "ctx");
        if (nbytes == 0) return;
        REQUIRE_PTR(dev, "dev");
        REQUIRE_PTR(host, "host");
        use_device(ctx);
        EMDEE_HIP_CHECK(hipMemcpyAsync(dev, host, nbytes, hipMemcpyHostToDevice, ctx->stream));
        EMDEE_HIP_CHECK(hipStreamSynchronize(ctx->stream));   // pageable host memory: keep it simple and safe
    });
}

int32_t emdee_memcpy_d2h(emdee_ctx *ctx, void *host, const void *dev, size_t nbytes) {
    return guarded([&] {
        REQUIRE_PTR(ctx, "ctx");
        if (nbytes == 0) return;
        REQUIRE_PTR(dev, "dev");
        REQUIRE_PTR(host, "host");
        use_device(ctx);
        EMDEE_HIP_CHECK(hipMemcpyAsync(host, dev, nbytes, hipMemcpyDeviceToHost, ctx->stream));
        EMDEE_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    });
}

int32_t emdee_memcpy_d2d(emdee_ctx *ctx, void *dst, const void *src, size_t nbytes) {
    return guarded([&] {
        REQUIRE_PTR(ctx, "ctx");
        if (nbytes == 0) return;
        REQUIRE_PTR(dst, "dst");
        REQUIRE_PTR(src, "src");
        use_device(ctx);
        EMDEE_HIP_CHECK(hipMemcpyAsync(dst, src, nbytes, hipMemcpyDeviceToDevice, ctx->stream));
    });
}

int32_t emdee_memset(emdee_ctx *ctx, void *dev, int32_t byte, size_t nbytes) {
    return guarded([&] {
        REQUIRE_PTR(ctx, "ctx");
        if (nbytes == 0) return;
        REQUIRE_PTR(dev, "dev");
        use_device(ctx);
        EMDEE_HIP_CHECK(hipMemsetAsync(dev, byte, nbytes, ctx->stream));
    });
}

// ---------------------------------------------------------------- pair function
int32_t emdee_interaction(emdee_ctx *ctx, int32_t n, const void *r2_dev, emdee_lj_model model, emdee_lj_atom atom_i,
                          emdee_lj_atom atom_j, int32_t mode, void *E_dev, void *W_dev, int32_t precision) {
    return guarded([&] {
        REQUIRE_PTR(ctx, "ctx");
        REQUIRE_PRECISION(precision);
        if (precision == EMDEE_F32) Factory<float>::interaction(ctx, n, r2_dev, model, atom_i, atom_j, mode, E_dev, W_dev);
        else Factory<double>::interaction(ctx, n, r2_dev, model, atom_i, atom_j, mode, E_dev, W_dev);
    });
}

// ---------------------------------------------------------------- Cells
int32_t emdee_cells_create(emdee_ctx *ctx, int32_t N, double L, double cutoff, int32_t ndiv, int32_t precision,
                           emdee_cells **out) {
    return guarded([&] {
        REQUIRE_PTR(ctx, "ctx");
        REQUIRE_PTR(out, "out");
        REQUIRE_PRECISION(precision);
        *out = nullptr;
        use_device(ctx);
        ICells *impl = precision == EMDEE_F32 ? Factory<float>::cells(ctx, N, L, cutoff, ndiv)
                                              : Factory<double>::cells(ctx, N, L, cutoff, ndiv);
        *out = new emdee_cells{ctx, impl};
    });
}
int32_t emdee_cells_update(emdee_cells *cells, const void *positions_dev) {
    return guarded([&] { REQUIRE_PTR(cells, "cells"); cells->impl->update(positions_dev); });
}
int32_t emdee_cells_destroy(emdee_cells *cells) {
    return guarded([&] {
        if (!cells) return;
        (void)hipSetDevice(cells->ctx->device);
        (void)hipStreamSynchronize(cells->ctx->stream);
        delete cells->impl;
        delete cells;
    });
}
int32_t emdee_cells_M(const emdee_cells *cells, int32_t *M) {
    return guarded([&] { REQUIRE_PTR(cells, "cells"); REQUIRE_PTR(M, "M"); *M = cells->impl->M(); });
}
int32_t emdee_cells_arrays(const emdee_cells *cells, const int32_t **index_dev, const int32_t **population_dev,
                           const int32_t **start_dev, const int32_t **order_dev) {
    return guarded([&] { REQUIRE_PTR(cells, "cells"); cells->impl->arrays(index_dev, population_dev, start_dev, order_dev); });
}

// ---------------------------------------------------------------- neighbour handle + operator
int32_t emdee_nbr_create(emdee_ctx *ctx, int32_t N, double skin, int32_t precision, emdee_nbr **out) {
    return guarded([&] {
        REQUIRE_PTR(ctx, "ctx");
        REQUIRE_PTR(out, "out");
        REQUIRE_PRECISION(precision);
        *out = nullptr;
        use_device(ctx);
        INbr *impl = precision == EMDEE_F32 ? Factory<float>::nbr(ctx, N, skin) : Factory<double>::nbr(ctx, N, skin);
        *out = new emdee_nbr{ctx, impl};
    });
}
int32_t emdee_nbr_destroy(emdee_nbr *nbr) {
    return guarded([&] {
        if (!nbr) return;
        (void)hipSetDevice(nbr->ctx->device);
        (void)hipStreamSynchronize(nbr->ctx->stream);
        delete nbr->impl;
        delete nbr;
    });
}
int32_t emdee_nbr_stats(emdee_nbr *nbr, int64_t *builds, int64_t *listed, int32_t *max_count, int32_t *capacity) {
    return guarded([&] { REQUIRE_PTR(nbr, "nbr"); nbr->impl->stats(builds, listed, max_count, capacity); });
}
int32_t emdee_nbr_count_pairs(emdee_nbr *nbr, int64_t *pairs_in_cutoff) {
    return guarded([&] { REQUIRE_PTR(nbr, "nbr"); REQUIRE_PTR(pairs_in_cutoff, "pairs_in_cutoff"); nbr->impl->count_pairs(pairs_in_cutoff); });
}

int32_t emdee_nbr_list(emdee_nbr *nbr, int32_t *counts_dev, int32_t *neighbors_dev, int32_t capacity) {
    return guarded([&] { REQUIRE_PTR(nbr, "nbr"); nbr->impl->export_list(counts_dev, neighbors_dev, capacity); });
}
int32_t emdee_md_nbr_list(emdee_md *md, int32_t *counts_dev, int32_t *neighbors_dev, int32_t capacity) {
    return guarded([&] { REQUIRE_PTR(md, "md"); md->impl->export_list(counts_dev, neighbors_dev, capacity); });
}

int32_t emdee_nbr_set_exclusions(emdee_nbr *nbr, const int32_t *pairs_dev, int32_t n_pairs) {
    return guarded([&] { REQUIRE_PTR(nbr, "nbr"); nbr->impl->set_pairs(pairs_dev, n_pairs, false, 1.0); });
}
int32_t emdee_nbr_set_pairs14(emdee_nbr *nbr, const int32_t *pairs_dev, int32_t n_pairs, double lj14scale) {
    return guarded([&] { REQUIRE_PTR(nbr, "nbr"); nbr->impl->set_pairs(pairs_dev, n_pairs, true, lj14scale); });
}
int32_t emdee_md_set_exclusions(emdee_md *md, const int32_t *pairs_dev, int32_t n_pairs) {
    return guarded([&] { REQUIRE_PTR(md, "md"); md->impl->set_pairs(pairs_dev, n_pairs, false, 1.0); });
}
int32_t emdee_md_set_pairs14(emdee_md *md, const int32_t *pairs_dev, int32_t n_pairs, double lj14scale) {
    return guarded([&] { REQUIRE_PTR(md, "md"); md->impl->set_pairs(pairs_dev, n_pairs, true, lj14scale); });
}

int32_t emdee_compute_nonbonded(emdee_ctx *ctx, void *forces_dev, void *energies_dev, void *virials_dev,
                                const void *positions_dev, double L, emdee_nbr *nbr, emdee_lj_model model,
                                const emdee_lj_atom *atoms_dev, int32_t bitmask, int32_t precision) {
    return guarded([&] {
        REQUIRE_PTR(ctx, "ctx");
        REQUIRE_PTR(nbr, "nbr (tiles)");
        REQUIRE_PRECISION(precision);
        EMDEE_REQUIRE(nbr->ctx == ctx, EMDEE_ERR_INVALID, "neighbour handle belongs to another context");
        // the handle was created for one precision; dynamic_cast-free check through the factory type
        nbr->impl->compute(forces_dev, energies_dev, virials_dev, positions_dev, L, model, atoms_dev, bitmask);
        (void)precision;
    });
}

int32_t emdee_compute_nonbonded_tiles(emdee_ctx *ctx, void *forces_dev, void *energies_dev, void *virials_dev,
                                      const void *positions_dev, double L, int32_t N, emdee_lj_model model,
                                      const emdee_lj_atom *atoms_dev, int32_t bitmask, int32_t mode, int32_t precision) {
    return guarded([&] {
        REQUIRE_PTR(ctx, "ctx");
        REQUIRE_PRECISION(precision);
        if (precision == EMDEE_F32)
            Factory<float>::tiles(ctx, forces_dev, energies_dev, virials_dev, positions_dev, L, N, model, atoms_dev, bitmask, mode);
        else
            Factory<double>::tiles(ctx, forces_dev, energies_dev, virials_dev, positions_dev, L, N, model, atoms_dev, bitmask, mode);
    });
}

int32_t emdee_compute_nonbonded_naive(emdee_ctx *ctx, void *forces_dev, void *energies_dev, void *virials_dev,
                                      const void *positions_dev, double L, int32_t N, emdee_lj_model model,
                                      const emdee_lj_atom *atoms_dev, int32_t mode, int32_t precision) {
    return guarded([&] {
        REQUIRE_PTR(ctx, "ctx");
        REQUIRE_PRECISION(precision);
        if (precision == EMDEE_F32)
            Factory<float>::naive(ctx, forces_dev, energies_dev, virials_dev, positions_dev, L, N, model, atoms_dev, mode);
        else
            Factory<double>::naive(ctx, forces_dev, energies_dev, virials_dev, positions_dev, L, N, model, atoms_dev, mode);
    });
}

// ---------------------------------------------------------------- velocity-Verlet
int32_t emdee_md_create(emdee_ctx *ctx, const double lo[3], const double len[3], const int32_t periodic[3],
                        emdee_lj_model model, double skin, int32_t precision, emdee_md **out) {
    return guarded([&] {
        REQUIRE_PTR(ctx, "ctx");
        REQUIRE_PTR(out, "out");
        REQUIRE_PTR(lo, "lo");
        REQUIRE_PTR(len, "len");
        REQUIRE_PTR(periodic, "periodic");
        REQUIRE_PRECISION(precision);
        *out = nullptr;
        use_device(ctx);
        IMd *impl = precision == EMDEE_F32 ? Factory<float>::md(ctx, lo, len, periodic, model, skin)
                                           : Factory<double>::md(ctx, lo, len, periodic, model, skin);
        *out = new emdee_md{ctx, impl};
    });
}
int32_t emdee_md_destroy(emdee_md *md) {
    return guarded([&] {
        if (!md) return;
        (void)hipSetDevice(md->ctx->device);
        (void)hipStreamSynchronize(md->ctx->stream);
        delete md->impl;
        delete md;
    });
}
int32_t emdee_md_set_state(emdee_md *md, int32_t n_owned, int32_t n_ghost, const void *positions_dev,
                           const void *velocities_dev, const emdee_lj_atom *atoms_dev, const void *inv_mass_dev) {
    return guarded([&] { REQUIRE_PTR(md, "md"); md->impl->set_state(n_owned, n_ghost, positions_dev, velocities_dev, atoms_dev, inv_mass_dev); });
}
int32_t emdee_md_get_state(emdee_md *md, void *positions_dev, void *velocities_dev, void *forces_dev, void *energies_dev,
                           void *virials_dev) {
    return guarded([&] { REQUIRE_PTR(md, "md"); md->impl->get_state(positions_dev, velocities_dev, forces_dev, energies_dev, virials_dev); });
}
int32_t emdee_md_step(emdee_md *md, int32_t nsteps, double dt, int32_t rebuild_every) {
    return guarded([&] { REQUIRE_PTR(md, "md"); md->impl->step(nsteps, dt, rebuild_every); });
}
int32_t emdee_md_kick_drift(emdee_md *md, double dt, double kick) {
    return guarded([&] { REQUIRE_PTR(md, "md"); md->impl->kick_drift(dt, kick); });
}
int32_t emdee_md_forces(emdee_md *md, int32_t bitmask, int32_t phase) {
    return guarded([&] {
        REQUIRE_PTR(md, "md");
        EMDEE_REQUIRE(phase >= 0 && phase <= 2, EMDEE_ERR_INVALID, "phase must be 0, 1 or 2");
        md->impl->forces(bitmask, phase);
    });
}
int32_t emdee_md_kick(emdee_md *md, double dt) {
    return guarded([&] { REQUIRE_PTR(md, "md"); md->impl->kick(dt); });
}
int32_t emdee_md_fused_step(emdee_md *md, double dt, double kick, int32_t phase, int32_t *fused) {
    return guarded([&] {
        REQUIRE_PTR(md, "md");
        EMDEE_REQUIRE(phase >= 0 && phase <= 2, EMDEE_ERR_INVALID, "phase must be 0, 1 or 2");
        const bool ok = md->impl->fused_step(dt, kick, phase);
        if (fused) *fused = ok ? 1 : 0;
    });
}
int32_t emdee_md_needs_rebuild(emdee_md *md, int32_t *flag) {
    return guarded([&] { REQUIRE_PTR(md, "md"); REQUIRE_PTR(flag, "flag"); *flag = md->impl->needs_rebuild() ? 1 : 0; });
}
int32_t emdee_md_rebuild(emdee_md *md) {
    return guarded([&] { REQUIRE_PTR(md, "md"); md->impl->rebuild(); });
}
int32_t emdee_md_pack_positions(emdee_md *md, const int32_t *ids_dev, const int32_t *codes_dev, int32_t n,
                                const double *shifts, int32_t n_shifts, void *buf_dev) {
    return guarded([&] {
        REQUIRE_PTR(md, "md");
        REQUIRE_PTR(shifts, "shifts");
        EMDEE_REQUIRE(n_shifts >= 1 && n_shifts <= 27, EMDEE_ERR_INVALID, "pack_positions: n_shifts must be in 1..27");
        EMDEE_REQUIRE(n == 0 || (ids_dev && buf_dev), EMDEE_ERR_INVALID, "pack_positions: NULL array");
        md->impl->pack_positions(ids_dev, codes_dev, n, shifts, n_shifts, buf_dev);
    });
}
int32_t emdee_md_unpack_ghosts(emdee_md *md, const void *buf_dev, int32_t first, int32_t n) {
    return guarded([&] {
        REQUIRE_PTR(md, "md");
        EMDEE_REQUIRE(n == 0 || buf_dev, EMDEE_ERR_INVALID, "unpack_ghosts: NULL buffer");
        md->impl->unpack_ghosts(buf_dev, first, n);
    });
}
int32_t emdee_md_energies(emdee_md *md, double out[3]) {
    return guarded([&] { REQUIRE_PTR(md, "md"); REQUIRE_PTR(out, "out"); md->impl->energies(out); });
}
int32_t emdee_md_nbr_stats(emdee_md *md, int64_t *builds, int64_t *listed, int32_t *max_count, int32_t *capacity) {
    return guarded([&] { REQUIRE_PTR(md, "md"); md->impl->stats(builds, listed, max_count, capacity); });
}
int32_t emdee_md_count_pairs(emdee_md *md, int64_t *pairs_in_cutoff) {
    return guarded([&] { REQUIRE_PTR(md, "md"); REQUIRE_PTR(pairs_in_cutoff, "pairs_in_cutoff"); md->impl->count_pairs(pairs_in_cutoff); });
}
int32_t emdee_md_profile(emdee_md *md, int32_t enable) {
    return guarded([&] { REQUIRE_PTR(md, "md"); md->impl->profile(enable != 0); });
}
int32_t emdee_md_kernel_time(emdee_md *md, int32_t kernel, double *total_ms, int64_t *launches) {
    return guarded([&] { REQUIRE_PTR(md, "md"); md->impl->kernel_time(kernel, total_ms, launches); });
}
int32_t emdee_md_set_langevin(emdee_md *md, double gamma, double temperature, uint64_t seed, uint64_t first_step) {
    return guarded([&] { REQUIRE_PTR(md, "md"); md->impl->set_langevin(gamma, temperature, seed, first_step); });
}
int32_t emdee_md_set_langevin_ids(emdee_md *md, const int64_t *ids_dev) {
    return guarded([&] { REQUIRE_PTR(md, "md"); md->impl->set_langevin_ids(ids_dev); });
}
int32_t emdee_md_langevin_normals(emdee_md *md, uint64_t seed, uint64_t step, const int64_t *ids_dev, int32_t n,
                                  double *out_dev) {
    return guarded([&] {
        REQUIRE_PTR(md, "md");
        EMDEE_REQUIRE(n >= 0 && (n == 0 || (ids_dev && out_dev)), EMDEE_ERR_INVALID, "langevin_normals: bad arguments");
        md->impl->langevin_normals(seed, step, ids_dev, n, out_dev);
    });
}

// ---------------------------------------------------------------- domain decomposition
int32_t emdee_dd_unique_id(uint8_t out[128]) {
    return guarded([&] {
        REQUIRE_PTR(out, "out");
        // resolved here (not through dd.hpp) so that this translation unit stays free of template code
        void *h = nullptr;
        const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char *n : names) {
            h = dlopen(n, RTLD_NOW | RTLD_LOCAL);
            if (h) break;
        }
        EMDEE_REQUIRE(h != nullptr, EMDEE_ERR_INVALID, "librccl.so.1 not found (%s)", dlerror());
        typedef struct { char internal[128]; } Id;
        auto get = reinterpret_cast<int (*)(Id *)>(dlsym(h, "ncclGetUniqueId"));
        EMDEE_REQUIRE(get != nullptr, EMDEE_ERR_INVALID, "librccl: missing symbol ncclGetUniqueId");
        Id id;
        const int r = get(&id);
        EMDEE_REQUIRE(r == 0, EMDEE_ERR_HIP, "ncclGetUniqueId failed with code %d", r);
        memcpy(out, &id, 128);
    });
}
int32_t emdee_dd_rccl_selftest(emdee_ctx *ctx, int32_t n_bytes) {
    return guarded([&] { REQUIRE_PTR(ctx, "ctx"); dd_rccl_selftest(ctx, n_bytes); });
}
int32_t emdee_dd_describe(const double len[3], const int32_t grid[3], double halo, int32_t rank, int32_t *ndirs,
                          int32_t dirs[78], int32_t dir_rank[26], double dir_shift[78], int32_t *npeers, int32_t peers[26],
                          double local_lo[3], double local_len[3], int32_t periodic[3]) {
    return guarded([&] {
        EMDEE_REQUIRE(len && grid && ndirs && dirs && dir_rank && dir_shift && npeers && peers && local_lo && local_len && periodic,
                      EMDEE_ERR_INVALID, "emdee_dd_describe: NULL argument");
        dd_describe(len, grid, halo, rank, ndirs, dirs, dir_rank, dir_shift, npeers, peers, local_lo, local_len, periodic);
    });
}
int32_t emdee_dd_create(emdee_ctx *ctx, const double len[3], const int32_t grid[3], int32_t rank_first, int32_t n_local,
                        const uint8_t *unique_id, emdee_lj_model model, double skin, int32_t precision, emdee_dd **out) {
    return guarded([&] {
        REQUIRE_PTR(ctx, "ctx");
        REQUIRE_PTR(out, "out");
        REQUIRE_PTR(len, "len");
        REQUIRE_PTR(grid, "grid");
        REQUIRE_PRECISION(precision);
        *out = nullptr;
        use_device(ctx);
        IDd *impl = precision == EMDEE_F32 ? Factory<float>::dd(ctx, len, grid, rank_first, n_local, unique_id, model, skin)
                                           : Factory<double>::dd(ctx, len, grid, rank_first, n_local, unique_id, model, skin);
        emdee_dd *dd = new emdee_dd{ctx, impl, {}};
        for (int l = 0; l < n_local; l++) dd->engines.push_back(new emdee_md{ctx, impl->engine(l)});
        *out = dd;
    });
}
int32_t emdee_dd_destroy(emdee_dd *dd) {
    return guarded([&] {
        if (!dd) return;
        (void)hipSetDevice(dd->ctx->device);
        for (emdee_md *m : dd->engines) delete m;          // borrowed views: the integrators belong to impl
        delete dd->impl;
        delete dd;
    });
}
int32_t emdee_dd_set_atoms(emdee_dd *dd, int32_t local, int32_t n, const void *positions_dev, const void *velocities_dev,
                           const emdee_lj_atom *atoms_dev, const int64_t *gids_dev) {
    return guarded([&] { REQUIRE_PTR(dd, "dd"); dd->impl->set_atoms(local, n, positions_dev, velocities_dev, atoms_dev, gids_dev); });
}
int32_t emdee_dd_load(emdee_dd *dd) {
    return guarded([&] { REQUIRE_PTR(dd, "dd"); dd->impl->load(); });
}
int32_t emdee_dd_step(emdee_dd *dd, int32_t nsteps, double dt, int32_t rebuild_every) {
    return guarded([&] { REQUIRE_PTR(dd, "dd"); dd->impl->step(nsteps, dt, rebuild_every); });
}
int32_t emdee_dd_energies(emdee_dd *dd, double out[3]) {
    return guarded([&] { REQUIRE_PTR(dd, "dd"); REQUIRE_PTR(out, "out"); dd->impl->energies(out); });
}
int32_t emdee_dd_counts(emdee_dd *dd, int32_t local, int64_t *n_global, int32_t *n_owned, int32_t *n_ghost) {
    return guarded([&] {
        REQUIRE_PTR(dd, "dd");
        if (n_global) *n_global = dd->impl->n_atoms_global();
        if (n_owned) *n_owned = dd->impl->n_owned(local);
        if (n_ghost) *n_ghost = dd->impl->n_ghost(local);
    });
}
int32_t emdee_dd_get_state(emdee_dd *dd, int32_t local, int64_t *gids_dev, void *positions_dev, void *velocities_dev,
                           void *forces_dev) {
    return guarded([&] { REQUIRE_PTR(dd, "dd"); dd->impl->get_state(local, gids_dev, positions_dev, velocities_dev, forces_dev); });
}
int32_t emdee_dd_engine(emdee_dd *dd, int32_t local, emdee_md **out) {
    return guarded([&] {
        REQUIRE_PTR(dd, "dd");
        REQUIRE_PTR(out, "out");
        EMDEE_REQUIRE(local >= 0 && local < (int32_t)dd->engines.size(), EMDEE_ERR_INVALID, "emdee_dd_engine: local domain out of range");
        *out = dd->engines[local];
    });
}
int32_t emdee_dd_set_langevin(emdee_dd *dd, double gamma, double temperature, uint64_t seed, uint64_t first_step) {
    return guarded([&] { REQUIRE_PTR(dd, "dd"); dd->impl->set_langevin(gamma, temperature, seed, first_step); });
}
int32_t emdee_dd_stats(emdee_dd *dd, int64_t out[4]) {
    return guarded([&] { REQUIRE_PTR(dd, "dd"); REQUIRE_PTR(out, "out"); dd->impl->stats(out); });
}
int32_t emdee_dd_rebuild_stats(emdee_dd *dd, int64_t out[4]) {
    return guarded([&] { REQUIRE_PTR(dd, "dd"); REQUIRE_PTR(out, "out"); dd->impl->rebuild_stats(out); });
}
int32_t emdee_dd_phase_times(emdee_dd *dd, double out[8]) {
    return guarded([&] { REQUIRE_PTR(dd, "dd"); REQUIRE_PTR(out, "out"); dd->impl->phase_times(out); });
}
int32_t emdee_dd_set_overlap(emdee_dd *dd, int32_t overlap) {
    return guarded([&] { REQUIRE_PTR(dd, "dd"); dd->impl->set_overlap(overlap != 0); });
}

}  // extern "C"

#ifdef EMDEE_BOUNDS
namespace emdee {
void bounds_poll_capi(int out[3]) { bounds_poll_here(out); }
}  // namespace emdee
#endif
