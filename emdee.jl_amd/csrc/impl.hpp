// impl.hpp -- the objects behind the C ABI handles, templated on the real type.
#pragma once

#include "allpairs.hpp"
#include "iface.hpp"
#include "nbsys.hpp"

namespace emdee {

// ------------------------------------------------------------------------------------ Cells
// Cells(r, L, cutoff; ndiv) / update_cells! -- src/cells.jl:176-222.  head/next linked lists become
// start/order arrays (counting sort); index and population keep the reference's meaning.
template <typename real>
struct CellsImpl : ICells {
    emdee_ctx *ctx;
    int N, Mdim;
    GridP<real> grid{};
    size_t ncell;
    DevBuf<int> index, population, start, fill, tmp, order;
    Scanner scanner;

    CellsImpl(emdee_ctx *c, int n, double L, double cutoff, int ndiv) : ctx(c), N(n) {
        EMDEE_REQUIRE(n >= 0 && L > 0 && cutoff > 0 && ndiv >= 1, EMDEE_ERR_INVALID, "Cells: need N >= 0, L > 0, cutoff > 0, ndiv >= 1");
        Mdim = std::max(1, (int)std::floor((double)ndiv * L / cutoff));   // src/cells.jl:36
        EMDEE_REQUIRE(Mdim <= 1290, EMDEE_ERR_INVALID, "Cells: M = %d cells per dimension overflows int32 ids", Mdim);
        for (int d = 0; d < 3; d++) {
            grid.lo[d] = 0; grid.len[d] = (real)L; grid.plen[d] = (real)L; grid.pinv[d] = (real)(1.0 / L);
            grid.per[d] = 1; grid.M[d] = Mdim;
        }
        grid.nd = ndiv;
        grid.one_based = 1;   // src/cells.jl:85,181
        ncell = (size_t)Mdim * Mdim * Mdim;
        index.ensure(n + 1); tmp.ensure(n + 1); order.ensure(n + 1);
        population.ensure(ncell + 2); start.ensure(ncell + 2); fill.ensure(ncell + 2);
    }

    void update(const void *positions) override {
        EMDEE_REQUIRE(positions || N == 0, EMDEE_ERR_INVALID, "Cells: positions is NULL");
        use_device(ctx);
        hipStream_t s = ctx->stream;
        EMDEE_HIP_CHECK(hipMemsetAsync(population.ptr, 0, (ncell + 1) * sizeof(int), s));
        EMDEE_HIP_CHECK(hipMemsetAsync(fill.ptr, 0, ncell * sizeof(int), s));
        if (N > 0)
            hipLaunchKernelGGL((k_cell_assign<real, UserPos<real>>), dim3(blocks_for(N, 256)), dim3(256), 0, s, N,
                               UserPos<real>{(const real *)positions}, grid, index.ptr, population.ptr);
        EMDEE_HIP_CHECK(hipMemcpyAsync(start.ptr, population.ptr, (ncell + 1) * sizeof(int), hipMemcpyDeviceToDevice, s));
        scanner.run(start.ptr, ncell + 1, s);
        if (N > 0) {
            hipLaunchKernelGGL(k_cell_scatter, dim3(blocks_for(N, 256)), dim3(256), 0, s, N, index.ptr, 1, start.ptr,
                               fill.ptr, tmp.ptr);
            hipLaunchKernelGGL(k_cell_rankfix, dim3(blocks_for(N, 256)), dim3(256), 0, s, N, index.ptr, 1, start.ptr,
                               tmp.ptr, (const int *)nullptr, order.ptr);
        }
        EMDEE_HIP_CHECK(hipGetLastError());
    }
    int M() const override { return Mdim; }
    void arrays(const int32_t **i, const int32_t **p, const int32_t **s, const int32_t **o) const override {
        if (i) *i = index.ptr;
        if (p) *p = population.ptr;
        if (s) *s = start.ptr;
        if (o) *o = order.ptr;
    }
};

// ------------------------------------------------------------------------------------ neighbour handle
template <typename real>
struct NbrImpl : INbr {
    NbSystem<real> sys;
    int N;
    double L_built = -1.0;

    NbrImpl(emdee_ctx *c, int n, double skin) : N(n) {
        EMDEE_REQUIRE(n >= 0 && skin >= 0, EMDEE_ERR_INVALID, "nbr: need N >= 0 and skin >= 0");
        sys.ctx = c;
        sys.skin = skin;
        // Float32 callers get the reference's Float32 pair geometry (EMDEE_F32_FAST=1: the MD loop's brick-relative tiles)
        sys.refmath = sizeof(real) == 4 && !std::getenv("EMDEE_F32_FAST");
    }

    void compute(void *forces, void *energies, void *virials, const void *positions, double L, const emdee_lj_model &model,
                 const emdee_lj_atom *atoms, int bitmask) override {
        use_device(sys.ctx);
        EMDEE_REQUIRE(bitmask >= 0 && bitmask <= 7, EMDEE_ERR_INVALID, "bitmask must be a combination of FORCES|ENERGIES|VIRIALS");
        EMDEE_REQUIRE(!(bitmask & EMDEE_FORCES) || forces || N == 0, EMDEE_ERR_INVALID, "forces selected but NULL");
        EMDEE_REQUIRE(!(bitmask & EMDEE_ENERGIES) || energies || N == 0, EMDEE_ERR_INVALID, "energies selected but NULL");
        EMDEE_REQUIRE(!(bitmask & EMDEE_VIRIALS) || virials || N == 0, EMDEE_ERR_INVALID, "virials selected but NULL");
        EMDEE_REQUIRE(L > 0, EMDEE_ERR_INVALID, "L must be positive");
        if (N == 0 || bitmask == 0) return;
        EMDEE_REQUIRE(positions && atoms, EMDEE_ERR_INVALID, "positions/atoms are NULL");
        const real *pos = (const real *)positions;
        const double lo[3] = {0, 0, 0}, len[3] = {L, L, L};
        const int per[3] = {1, 1, 1};   // cubic periodic box, the reference's only geometry (Q9)
        sys.set_box(lo, len, per);
        sys.set_model(model, sys.skin);
        bool rebuild = !sys.has_list || sys.n_total != N;
        // list kept: one pass refreshes the records, tests the displacements and re-checks the species (the caller
        // may have edited atoms); one read-back
        bool resorted = false;
        if (!rebuild) {
            const bool was_uniform = sys.uniform_atoms;
            rebuild = sys.refresh_and_check(pos, atoms);
            // The list is outrun, but the pass above has just written the caller's positions (and LJAtom values) into the
            // cell-ordered records: an untyped box whose kernel class has not changed re-sorts from THOSE, as an MD rebuild does
            // -- nearly sorted input (one integer atomic per run of equal cells instead of one per atom), no second look at the
            // LJAtom array (k_atoms_differ, k_species_collect and their read-back).  Round 4: a reload 4.3 -> 3.7 ms at 10^7
            // atoms.  (Typed boxes are sorted by species, which an edited LJAtom array invalidates: they load afresh.)
            if (rebuild && sys.sorted && sys.nt == 1 && sys.uniform_atoms == was_uniform && !std::getenv("EMDEE_OPERATOR_RELOAD")) {
                sys.resort();
                resorted = true;
            }
        }
        if (rebuild && !resorted) sys.load_user(N, 0, pos, nullptr, atoms, nullptr);
        real *uf = (bitmask & EMDEE_FORCES) ? (real *)forces : nullptr, *ue = (bitmask & EMDEE_ENERGIES) ? (real *)energies : nullptr,
             *uw = (bitmask & EMDEE_VIRIALS) ? (real *)virials : nullptr;
        if (sys.brick_active) {
            // the tiled kernels write the caller's arrays themselves (owner lane, caller index from perm)
            sys.out_f = uf; sys.out_e = ue; sys.out_w = uw;
            sys.ref_pos = pos;
            sys.compute_forces(bitmask);
            sys.ref_pos = nullptr;
            sys.out_f = sys.out_e = sys.out_w = nullptr;
        } else {
            sys.compute_forces(bitmask);
            sys.unsort(nullptr, nullptr, uf, ue, uw);
        }
        EMDEE_HIP_CHECK(hipGetLastError());
    }
    void stats(int64_t *builds, int64_t *listed, int32_t *max_count, int32_t *capacity) override {
        use_device(sys.ctx);
        sys.list_stats(false, listed, max_count, nullptr);
        if (builds) *builds = sys.builds;
        if (capacity) *capacity = sys.stride;
    }
    void count_pairs(int64_t *pairs) override {
        use_device(sys.ctx);
        sys.list_stats(true, nullptr, nullptr, pairs);
    }
    void export_list(int32_t *counts, int32_t *neighbors, int32_t capacity) override {
        use_device(sys.ctx);
        sys.export_list(counts, neighbors, capacity);
    }
    void set_pairs(const int32_t *pairs, int32_t n_pairs, bool one_four, double lj14scale) override {
        use_device(sys.ctx);
        sys.set_pair_tables(N, one_four ? nullptr : pairs, one_four ? 0 : n_pairs, !one_four, one_four ? pairs : nullptr, one_four ? n_pairs : 0, one_four, lj14scale);
    }
};

// ------------------------------------------------------------------------------------ velocity-Verlet
template <typename real>
struct MdImpl : IMd {
    NbSystem<real> sys;
    int n_ghost = 0;
    int since_build = 0;
    int current_mask = 0;   // which of f/e/w match the current positions

    MdImpl(emdee_ctx *c, const double lo[3], const double len[3], const int32_t per[3], const emdee_lj_model &model,
           double skin) {
        sys.ctx = c;
        int p[3] = {per[0], per[1], per[2]};
        sys.set_box(lo, len, p);
        sys.set_model(model, skin);
        sys.with_vel = true;
    }

    void set_state(int n_owned, int ng, const void *pos, const void *vel, const emdee_lj_atom *atoms,
                   const void *inv_mass) override {
        use_device(sys.ctx);
        EMDEE_REQUIRE(vel || n_owned == 0, EMDEE_ERR_INVALID, "velocities are NULL");
        n_ghost = ng;
        sys.load_user(n_owned, ng, (const real *)pos, (const real *)vel, atoms, (const real *)inv_mass, tags_user);
        since_build = 0;
        current_mask = 0;
        if (!defer_forces) {
            sys.compute_forces(EMDEE_FORCES);
            current_mask = EMDEE_FORCES;
        }
        EMDEE_HIP_CHECK(hipGetLastError());
    }
    // emdee_dd_step: the rebuild in the middle of a run is followed by a fused step, which evaluates the forces itself
    bool defer_forces = false;
    // decomposed domains: the global ids of the atoms handed to set_state (caller order, owned atoms and ghosts); they travel
    // with the atoms from then on (NbSystem::tag) and order the atoms of a cell
    const long long *tags_user = nullptr;
    // an engine of an in-process decomposition runs on a stream of the library's own: the context its queries answer to
    // (common.hpp FenceOut; NULL: the engine's context is the caller's)
    const emdee_ctx *caller_ctx = nullptr;
    void get_state(void *pos, void *vel, void *frc, void *en, void *vir) override {
        use_device(sys.ctx);
        EMDEE_REQUIRE(sys.sorted, EMDEE_ERR_STATE, "md: no state loaded");
        if ((en || vir) && (current_mask & 6) != 6) forces(7, 0);
        sys.ids_map();                                       // (scratch of the engine's own: before the fence)
        FenceOut fence(caller_ctx, sys.stream());
        sys.unsort((real *)pos, (real *)vel, (real *)frc, (real *)en, (real *)vir);
        EMDEE_HIP_CHECK(hipGetLastError());
    }
    void step(int nsteps, double dt, int rebuild_every) override {
        use_device(sys.ctx);
        EMDEE_REQUIRE(sys.sorted, EMDEE_ERR_STATE, "md: no state loaded");
        EMDEE_REQUIRE(n_ghost == 0, EMDEE_ERR_STATE, "md_step needs n_ghost == 0; decomposed runs drive kick_drift/forces/kick");
        EMDEE_REQUIRE(nsteps >= 0 && dt >= 0, EMDEE_ERR_INVALID, "md_step: negative nsteps or dt");
        if (nsteps == 0) return;
        if (!(current_mask & EMDEE_FORCES)) forces(EMDEE_FORCES, 0);
        // x_1 = x_0 + dt (v_0 + dt/2 f_0); then every inner step is ONE kernel (force + full kick + drift:
        // the closing half kick of step s rides on the opening half kick of step s+1); the last step ends
        // with a plain force pass and the closing half kick.
        sys.kick_drift(0.5 * dt, dt);
        int s = 1;
        if (rebuild_every == 0) {
            // displacement-triggered rebuilds: the inner steps are queued a few at a time (one read-back per
            // batch, NbSystem::fused_steps_run_ahead); same sequence of states as one step at a time
            bool stale = sys.read_rebuild_flag();
            while (s < nsteps) {
                if (stale) { sys.resort(); since_build = 0; }
                const int ran = sys.fused_steps_run_ahead(dt, dt, nsteps - s, &stale);
                if (ran == 0) break;                          // direct kernels: one step at a time below
                s += ran; since_build += ran;
            }
            if (s == nsteps) {
                since_build++;
                if (stale) { sys.resort(); since_build = 0; }
                sys.compute_forces(EMDEE_FORCES);
                s++;
            }
        }
        for (; s <= nsteps; s++) {
            since_build++;
            bool rb = rebuild_every > 0 ? since_build >= rebuild_every : sys.read_rebuild_flag();
            if (rb) { sys.resort(); since_build = 0; }
            if (s == nsteps) {
                sys.compute_forces(EMDEE_FORCES);
            } else if (!sys.fused_step(dt, dt, 0)) {
                sys.compute_forces(EMDEE_FORCES);
                sys.kick_drift(dt, dt);
            }
        }
        sys.kick(0.5 * dt);
        current_mask = EMDEE_FORCES;
        EMDEE_HIP_CHECK(hipGetLastError());
    }
    void kick_drift(double dt, double kick) override {
        use_device(sys.ctx);
        sys.kick_drift(kick * dt, dt);
        since_build++;
        current_mask = 0;
    }
    void forces(int bitmask, int phase = 0) override {
        use_device(sys.ctx);
        sys.compute_forces(bitmask, phase);
        current_mask = phase == 1 ? 0 : bitmask;
        EMDEE_HIP_CHECK(hipGetLastError());
    }
    void kick(double dt) override {
        use_device(sys.ctx);
        sys.kick(0.5 * dt);
    }
    bool fused_step(double dt, double kick, int phase) override {
        use_device(sys.ctx);
        const bool ok = sys.fused_step(kick * dt, dt, phase);
        if (ok) { since_build += (phase != 1) ? 1 : 0; current_mask = 0; }
        EMDEE_HIP_CHECK(hipGetLastError());
        return ok;
    }
    bool needs_rebuild() override {
        use_device(sys.ctx);
        return sys.read_rebuild_flag();
    }
    void rebuild() override {
        use_device(sys.ctx);
        sys.resort();
        since_build = 0;
        current_mask = 0;
        EMDEE_HIP_CHECK(hipGetLastError());
    }
    void pack_positions(const int32_t *ids, const int32_t *codes, int n, const double *shifts, int n_shifts,
                        void *buf) override {
        use_device(sys.ctx);
        EMDEE_REQUIRE(sys.sorted, EMDEE_ERR_STATE, "md: no state loaded");
        if (n <= 0) return;
        ShiftTable<real> tab{};
        for (int k = 0; k < n_shifts; k++)
            for (int d = 0; d < 3; d++) tab.s[k][d] = (real)shifts[3 * k + d];
        hipLaunchKernelGGL((k_pack_positions<real>), dim3(blocks_for(n, 256)), dim3(256), 0, sys.stream(), n, ids, codes,
                           n_shifts, sys.inv_perm.ptr, sys.rec.ptr, tab, (real *)buf);
    }
    void unpack_ghosts(const void *buf, int first, int n) override {
        use_device(sys.ctx);
        EMDEE_REQUIRE(sys.sorted, EMDEE_ERR_STATE, "md: no state loaded");
        EMDEE_REQUIRE(first >= 0 && n >= 0 && first + n <= n_ghost, EMDEE_ERR_INVALID, "unpack_ghosts: range outside the ghosts");
        if (n == 0) return;
        hipLaunchKernelGGL((k_unpack_ghosts<real>), dim3(blocks_for(n, 256)), dim3(256), 0, sys.stream(), n,
                           sys.n_owned + first, sys.inv_perm.ptr, (const real *)buf, sys.rec.ptr);
        current_mask = 0;
    }
    void energies(double out[3]) override {
        use_device(sys.ctx);
        EMDEE_REQUIRE(sys.sorted, EMDEE_ERR_STATE, "md: no state loaded");
        if ((current_mask & 7) != 7) forces(7, 0);
        sys.energy_sums(0.0, out);
    }
    void stats(int64_t *builds, int64_t *listed, int32_t *max_count, int32_t *capacity) override {
        use_device(sys.ctx);
        sys.list_stats(false, listed, max_count, nullptr);
        if (builds) *builds = sys.builds;
        if (capacity) *capacity = sys.stride;
    }
    void count_pairs(int64_t *pairs) override {
        use_device(sys.ctx);
        sys.list_stats(true, nullptr, nullptr, pairs);
    }
    void export_list(int32_t *counts, int32_t *neighbors, int32_t capacity) override {
        use_device(sys.ctx);
        sys.ids_map();
        FenceOut fence(caller_ctx, sys.stream());
        sys.export_list(counts, neighbors, capacity);
    }
    void profile(bool enable) override {
        sys.profiling = enable;
        for (auto &t : sys.timers) t.reset();
    }
    void kernel_time(int kernel, double *total_ms, int64_t *launches) override {
        // ids 0..3: the TimerIds; 4: every fused step launch (interior + boundary halves of a decomposed step together, as
        // before they had timers of their own); 5: all but the boundary halves; 6: the boundary halves; 7: the halo of a
        // decomposed step (pack -> exchange -> unpack)
        EMDEE_REQUIRE(kernel >= 0 && kernel <= 7, EMDEE_ERR_INVALID, "kernel id out of range");
        use_device(sys.ctx);
        const int ids[8][2] = {{T_FORCE, -1}, {T_KICK_DRIFT, -1}, {T_REBUILD, -1}, {T_KICK, -1}, {T_STEP, T_STEP_BOUNDARY}, {T_STEP, -1},
                               {T_STEP_BOUNDARY, -1}, {T_HALO, -1}};
        double ms = 0.0;
        int64_t n = 0;
        for (int q = 0; q < 2; q++) {
            if (ids[kernel][q] < 0) continue;
            sys.timers[ids[kernel][q]].collect();
            ms += sys.timers[ids[kernel][q]].total_ms;
            n += sys.timers[ids[kernel][q]].launches;
        }
        if (total_ms) *total_ms = ms;
        if (launches) *launches = n;
    }
    void set_langevin(double gamma, double temperature, uint64_t seed, uint64_t first_step) override {
        sys.set_langevin(gamma, temperature, seed, first_step, sys.lgv_ids);
    }
    void set_langevin_ids(const int64_t *ids) override { sys.lgv_ids = reinterpret_cast<const long long *>(ids); }
    void set_pairs(const int32_t *pairs, int32_t n_pairs, bool one_four, double lj14scale) override {
        use_device(sys.ctx);
        EMDEE_REQUIRE(sys.sorted && n_ghost == 0 && !sys.id_gaps, EMDEE_ERR_STATE, "exclusions / 1-4 pairs: set them on a loaded integrator without ghosts (call emdee_md_set_state first)");
        sys.set_pair_tables(sys.n_owned, one_four ? nullptr : pairs, one_four ? 0 : n_pairs, !one_four, one_four ? pairs : nullptr, one_four ? n_pairs : 0, one_four, lj14scale);
        sys.resort();                                        // the list without the named pairs (a two-species box leaves the typed kernels)
        since_build = 0;
        sys.compute_forces(EMDEE_FORCES);
        current_mask = EMDEE_FORCES;
        EMDEE_HIP_CHECK(hipGetLastError());
    }
    void langevin_normals(uint64_t seed, uint64_t step, const int64_t *ids, int n, double *out) override {
        use_device(sys.ctx);
        if (n <= 0) return;
        hipLaunchKernelGGL((k_langevin_normals_test<real>), dim3(blocks_for(n, 256)), dim3(256), 0, sys.stream(), n,
                           (unsigned long long)seed, (unsigned long long)step, reinterpret_cast<const long long *>(ids), out);
        EMDEE_HIP_CHECK(hipGetLastError());
    }
};

// ------------------------------------------------------------------------------------ factories
template <typename real>
ICells *Factory<real>::cells(emdee_ctx *ctx, int N, double L, double cutoff, int ndiv) {
    return new CellsImpl<real>(ctx, N, L, cutoff, ndiv);
}
template <typename real>
INbr *Factory<real>::nbr(emdee_ctx *ctx, int N, double skin) {
    return new NbrImpl<real>(ctx, N, skin);
}
template <typename real>
IMd *Factory<real>::md(emdee_ctx *ctx, const double lo[3], const double len[3], const int32_t per[3],
                       const emdee_lj_model &model, double skin) {
    return new MdImpl<real>(ctx, lo, len, per, model, skin);
}

template <typename real>
void Factory<real>::tiles(emdee_ctx *ctx, void *f, void *e, void *w, const void *pos, double L, int N,
                          const emdee_lj_model &model, const emdee_lj_atom *atoms, int bitmask, int mode) {
    use_device(ctx);
    EMDEE_REQUIRE(N >= 0 && L > 0, EMDEE_ERR_INVALID, "tiles: need N >= 0 and L > 0");
    EMDEE_REQUIRE(bitmask >= 0 && bitmask <= 7, EMDEE_ERR_INVALID, "bitmask must be a combination of FORCES|ENERGIES|VIRIALS");
    EMDEE_REQUIRE(mode == EMDEE_LITERAL || mode == EMDEE_CUTOFF, EMDEE_ERR_INVALID, "mode must be LITERAL or CUTOFF");
    if (N == 0 || bitmask == 0) return;
    EMDEE_REQUIRE(pos && atoms, EMDEE_ERR_INVALID, "positions/atoms are NULL");
    EMDEE_REQUIRE((!(bitmask & 1) || f) && (!(bitmask & 2) || e) && (!(bitmask & 4) || w), EMDEE_ERR_INVALID,
                  "a selected output is NULL");
    LJModel<real> m = make_model<real>(model);
    const int nt = (N + TILE - 1) / TILE;
    if (mode == EMDEE_LITERAL)
        hipLaunchKernelGGL((k_tiles<real, EMDEE_LITERAL>), dim3(nt), dim3(TILE_BLOCK), 0, ctx->stream, N, (const real *)pos,
                           (real)L, atoms, m, bitmask, (real *)f, (real *)e, (real *)w);
    else
        hipLaunchKernelGGL((k_tiles<real, EMDEE_CUTOFF>), dim3(nt), dim3(TILE_BLOCK), 0, ctx->stream, N, (const real *)pos,
                           (real)L, atoms, m, bitmask, (real *)f, (real *)e, (real *)w);
    EMDEE_HIP_CHECK(hipGetLastError());
}

template <typename real>
void Factory<real>::naive(emdee_ctx *ctx, void *f, void *e, void *w, const void *pos, double L, int N,
                          const emdee_lj_model &model, const emdee_lj_atom *atoms, int mode) {
    use_device(ctx);
    EMDEE_REQUIRE(N >= 0 && L > 0, EMDEE_ERR_INVALID, "naive: need N >= 0 and L > 0");
    EMDEE_REQUIRE(mode == EMDEE_LITERAL || mode == EMDEE_CUTOFF, EMDEE_ERR_INVALID, "mode must be LITERAL or CUTOFF");
    if (N == 0) return;
    EMDEE_REQUIRE(pos && atoms && f && e && w, EMDEE_ERR_INVALID, "naive: NULL array");
    LJModel<real> m = make_model<real>(model);
    if (mode == EMDEE_LITERAL)
        hipLaunchKernelGGL((k_naive<real, EMDEE_LITERAL>), dim3(blocks_for(N, 64)), dim3(64), 0, ctx->stream, N,
                           (const real *)pos, (real)L, atoms, m, (real *)f, (real *)e, (real *)w);
    else
        hipLaunchKernelGGL((k_naive<real, EMDEE_CUTOFF>), dim3(blocks_for(N, 64)), dim3(64), 0, ctx->stream, N,
                           (const real *)pos, (real)L, atoms, m, (real *)f, (real *)e, (real *)w);
    EMDEE_HIP_CHECK(hipGetLastError());
}

template <typename real>
void Factory<real>::interaction(emdee_ctx *ctx, int n, const void *r2, const emdee_lj_model &model, emdee_lj_atom ai,
                                emdee_lj_atom aj, int mode, void *E, void *W) {
    use_device(ctx);
    EMDEE_REQUIRE(n >= 0, EMDEE_ERR_INVALID, "interaction: negative n");
    if (n == 0) return;
    EMDEE_REQUIRE(r2 && E && W, EMDEE_ERR_INVALID, "interaction: NULL array");
    hipLaunchKernelGGL((k_interaction<real>), dim3(blocks_for(n, 256)), dim3(256), 0, ctx->stream, n, (const real *)r2,
                       make_model<real>(model), ai, aj, mode, (real *)E, (real *)W);
    EMDEE_HIP_CHECK(hipGetLastError());
}

}  // namespace emdee
