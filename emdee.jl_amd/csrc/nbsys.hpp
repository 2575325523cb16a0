// nbsys.hpp -- host-side driver of the cell-ordered system: owns the HBM buffers and enqueues
// the kernels of kernels.hpp / brick.hpp on the context's stream.  Instantiated for float and double.
#pragma once
#include <cstdio>

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <type_traits>

#include "brick.hpp"
#ifdef EMDEE_EXPERIMENTS
#include "brick_tbuild.hpp"
#endif
#include "kernels.hpp"
#include "typed.hpp"

namespace emdee {

// The laboratory of rounds 1-5 -- measured alternatives that lost and the ablation switches that measured them -- compiles
// only under `make EXPERIMENTS=1` (-> libemdee_hip_exp.so).  The product library carries none of it: no experiment kernel is
// instantiated, and an experiment switch found in the environment is REFUSED with a message when an engine is created, not
// ignored (a run that believes it is measuring a variant must not silently measure the default).
#ifdef EMDEE_EXPERIMENTS
static inline const char *exp_env(const char *name) { return std::getenv(name); }
static inline void refuse_experiment_switches() {}
#else
static inline const char *exp_env(const char *) { return nullptr; }
static inline void refuse_experiment_switches() {
    static const char *const gone[] = {"EMDEE_DEBUG_RC2_SCALE", "EMDEE_TBUILD", "EMDEE_BUILD4", "EMDEE_BUILD_ALG", "EMDEE_BUILD_CHUNKED",
        "EMDEE_BUILD_STRIDED", "EMDEE_NO_BRICK_TABLES", "EMDEE_NO_PREMUL", "EMDEE_BUILD_NEARFAR", "EMDEE_NEAR_DELTA", "EMDEE_FAR_SKIP",
        "EMDEE_PLAN_MAXIMA", "EMDEE_PLAN_SYNC", "EMDEE_STRIDE", "EMDEE_BRICK_VARIANT"};
    for (const char *name : gone)
        EMDEE_REQUIRE(std::getenv(name) == nullptr, EMDEE_ERR_INVALID,
                      "%s is an experiment switch: this libemdee_hip.so was built without them (make -C emdee.jl_amd/csrc EXPERIMENTS=1 builds "
                      "libemdee_hip_exp.so, selected with EMDEE_HIP_LIB)", name);
}
#endif

// T_STEP: every fused step launch but the boundary-brick halves of a decomposed step, which go to T_STEP_BOUNDARY (emdee_md_kernel_time(4)
// reports the two together, 5 and 6 one each); T_HALO: pack -> exchange -> unpack of a decomposed step, on the stream they run on
enum TimerId { T_FORCE = 0, T_KICK_DRIFT = 1, T_REBUILD = 2, T_KICK = 3, T_STEP = 4, T_STEP_BOUNDARY = 5, T_HALO = 6, T_COUNT = 7 };
enum PathId { PATH_BRICK = 0, PATH_DIRECT = 1 };

// in-place exclusive scan of int32 data[0..n) (n may exceed one tile: recursive tile sums)
struct Scanner {
    DevBuf<int> scratch;
    void run(int *data, size_t n, hipStream_t s) {
        if (n == 0) return;
        if (n <= (size_t)SCAN_BLOCK_MAX) {
            hipLaunchKernelGGL(k_scan_block, dim3(1), dim3(1024), 0, s, data, (int)n);
            return;
        }
        size_t total = 0;
        for (size_t m = n; m > 1;) {
            m = (m + SCAN_TILE - 1) / SCAN_TILE;
            total += m;
            if (m == 1) break;
        }
        scratch.ensure(total + 1);
        level(data, n, scratch.ptr, s);
    }

  private:
    void level(int *data, size_t n, int *sums, hipStream_t s) {
        size_t tiles = (n + SCAN_TILE - 1) / SCAN_TILE;
        hipLaunchKernelGGL(k_scan_tiles, dim3((unsigned)tiles), dim3(SCAN_THREADS), 0, s, data, data, n,
                           tiles > 1 ? sums : nullptr);
        if (tiles > 1) {
            level(sums, tiles, sums + tiles, s);
            hipLaunchKernelGGL(k_scan_add, dim3(blocks_for(n, 256)), dim3(256), 0, s, data, n, sums);
        }
    }
};

// ---- brick kernel variants (shape, workgroup size, lanes per atom); variant 0 is the default ----
constexpr int BRICK_VARIANTS = 10;
constexpr size_t LDS_LIMIT = 160 * 1024;

template <int V>
struct BrickVariant;
// G = lanes per atom in the force kernels (and the row layout), GB = lanes per atom in the build kernel
template <> struct BrickVariant<0> { using Shape = BrickShape<4, 2, 2>; static constexpr int THREADS = 512, G = 4, GB = 8; };
template <> struct BrickVariant<1> { using Shape = BrickShape<3, 2, 2>; static constexpr int THREADS = 512, G = 8, GB = G; };
template <> struct BrickVariant<2> { using Shape = BrickShape<5, 2, 2>; static constexpr int THREADS = 512, G = 8, GB = G; };
template <> struct BrickVariant<3> { using Shape = BrickShape<4, 2, 2>; static constexpr int THREADS = 512, G = 16, GB = G; };
template <> struct BrickVariant<4> { using Shape = BrickShape<6, 2, 2>; static constexpr int THREADS = 512, G = 8, GB = G; };
template <> struct BrickVariant<5> { using Shape = BrickShape<2, 2, 2>; static constexpr int THREADS = 256, G = 8, GB = G; };
template <> struct BrickVariant<6> { using Shape = BrickShape<3, 3, 2>; static constexpr int THREADS = 512, G = 8, GB = G; };
// (7: the 1024-thread workgroup of variant 8 with FOUR lanes per atom: long rows, e.g. the rc = 3.5 sigma mixture -- 184 entries
// are 46 pair steps per lane -- amortise a round's fixed work over 16 atoms per wavefront instead of 8)
template <> struct BrickVariant<7> { using Shape = BrickShape<4, 2, 2>; static constexpr int THREADS = 1024, G = 4, GB = 8; };
template <> struct BrickVariant<8> { using Shape = BrickShape<4, 2, 2>; static constexpr int THREADS = 1024, G = 8, GB = G; };
// (9, round 5: two-species boxes with long rows on bricks of 2 x 2 x 2 cells -- a tile of 64 cells, ~2850 records at rc = 3.5 sigma,
// fits TWO 512-thread workgroups on a CU where variant 7's 96-cell tile leaves room for one of 1024: while one workgroup stages
// its tile or is being replaced, the other computes)
template <> struct BrickVariant<9> { using Shape = BrickShape<2, 2, 2>; static constexpr int THREADS = 512, G = 4, GB = 8; };

template <class F>
static inline void with_brick_variant(int v, F &&f) {
    switch (v) {
#ifdef EMDEE_EXPERIMENTS                                   // (brick shapes / lanes per atom of the tuning sweeps, EMDEE_BRICK_VARIANT)
        case 1: f(BrickVariant<1>{}); break;
        case 2: f(BrickVariant<2>{}); break;
        case 3: f(BrickVariant<3>{}); break;
        case 4: f(BrickVariant<4>{}); break;
        case 5: f(BrickVariant<5>{}); break;
        case 6: f(BrickVariant<6>{}); break;
#endif
        case 7: f(BrickVariant<7>{}); break;
        case 8: f(BrickVariant<8>{}); break;
        case 9: f(BrickVariant<9>{}); break;
        default: f(BrickVariant<0>{}); break;
    }
}

template <typename K>
static inline void allow_big_lds(K kernel, size_t bytes) {
    if (bytes > 48 * 1024)
        EMDEE_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(kernel),
                                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
}

template <typename real>
struct NbSystem {
    emdee_ctx *ctx = nullptr;
    double lo[3] = {0, 0, 0}, len[3] = {0, 0, 0};
    int per[3] = {1, 1, 1};
    double skin = 0.3, rlist = 0.0;
    emdee_lj_model model_d{};
    LJModel<real> model{};
    GridP<real> grid{};
    size_t ncell = 0;
    int n_total = 0, n_owned = 0;
    size_t pitch = 0;
    bool with_vel = false, with_mass = false;
    bool sorted = false, has_list = false;
    int stride = 0;
    int64_t builds = 0;
    bool profiling = false;
    KernelTimer timers[T_COUNT];

    // path selection
    int path = PATH_BRICK;
    int variant = 0;
    bool variant_forced = false;          // EMDEE_BRICK_VARIANT given: no automatic choice
    BrickGrid bgrid{};
    int tile_cap = 0, own_cap = 0, row_block = 1;
    int build_alg = 1;                    // k_brick_build ALG (2 = two-phase; 1 when a tile row is too crowded for it)
    bool force_build1 = false;            // EMDEE_BUILD_ALG=1: A/B switch
    int build_alg_pref = 3;               // EMDEE_BUILD_ALG=2: the two-phase build with the per-lane candidate loop
    // the transposed build (brick_tbuild.hpp: candidates in the lanes, own atoms in the loop) for the default variant of untyped
    // boxes.  Measured in round 4 and left OFF (EMDEE_TBUILD=1 switches it on): same neighbour set, 2.63 ms per build at 10^7
    // atoms against k_brick_build's 1.90 (profiles/r04/tbuild_*.txt).  tbuild_blocked: a cell of this state has more candidates
    // than the kernel's registers hold (it raised flags[3]) -- until the next load
    bool tbuild_enabled = exp_env("EMDEE_TBUILD") != nullptr && std::atoi(exp_env("EMDEE_TBUILD")) != 0 &&
                          exp_env("EMDEE_BUILD_ALG") == nullptr && exp_env("EMDEE_BUILD_NEARFAR") == nullptr;
    bool tbuild_blocked = false;
    // Four lanes per atom in the round-robin build (round 4; EMDEE_BUILD4=1 switches it on): 16 atoms per wavefront share what
    // is paid once per atom and wavefront-round (row bookkeeping, trip counts, prefix: 400 of ~960 instructions per atom and lane
    // at 8 lanes), the emission loops run on twice the hits per lane (less imbalance), and twice the row buffers fit because the
    // workgroup is 768 threads, two per CU: still six waves per SIMD.  Measured: 1.927 ms per launch against 1.90-1.93 with 8
    // lanes and 512 threads -- 11 % fewer vector instructions (SQ counters) and no gain: in both forms the waves sit at
    // s_waitcnt for 42-48 % of their lifetime, and doubling the candidate LDS reads costs only 12 % (profiles/r04/
    // tbuild_transposed_build.txt): dependent-chain latency, which the dealing does not change.  Off by default.  A lane's share of a tile row must fit a 16-bit field (64 slots per row): the x sub-bins
    // see to that in a fluid; the kernel reports a wider row in flags[4] and the state goes on with 8 lanes.
    bool build4_enabled = exp_env("EMDEE_BUILD4") != nullptr && std::atoi(exp_env("EMDEE_BUILD4")) != 0;
    bool build4_blocked = false;
    bool field16_blocked = false;         // a 16-bit hit field overflowed under a plan that should have ruled it out: 32-bit fields until the next load
    static constexpr int B4_THREADS = 768, B4_G = 4;
    static constexpr int TB_NPAIR = 5;    // 640 candidates per own cell (27 cells of 17.6 atoms at rho* = 0.8, r_list = 2.8: 475)
    template <class V>
    bool tbuild_active() const {
#ifdef EMDEE_EXPERIMENTS
        return tbuild_enabled && !tbuild_blocked && !typed_active && std::is_same<V, BrickVariant<0>>::value;
#else
        return false;
#endif
    }
    size_t lds_bytes = 0, lds_build_bytes = 0;
    float build_margin = 0.f;

    DevBuf<Rec<real>> rec, rec2;
    DevBuf<float> te, te2;
    DevBuf<real> vel, vel2, frc, en, vir, im, im2, xb, noise;
    DevBuf<int> perm, perm2, inv_perm, cell_of, cell_sorted, cell_sorted2, order, count, fill, nbr, cnt, flags, img, img2;
    DevBuf<int2> tmp2;                    // {id, sort key} in arrival order inside each cell (bin)
    // 64-bit ids that travel with the atoms through every sort (decomposed domains: the GLOBAL ids, owned atoms and ghosts).
    // With them the order inside a cell is by the low word of the tag, not by the local id: the cell order -- and with it the
    // order of every neighbour row and force sum -- no longer depends on how a domain numbers its atoms, i.e. on the history
    // of migrations or on the path a rebuild took (counted, count-free, in the engine's own order).
    DevBuf<long long> tag, tag2;
    bool use_tags = false;
    size_t cap_hint = 0;                  // reserve() sizes the per-atom arrays (and the plane pitch) for at least this many slots
    size_t capacity = 0;                  // ... what they hold
    bool has_ghosts = false;
    // ids: perm[p] < n_owned <=> owned.  After resort_edit the ids in use have gaps (id_space > n_total); callers that want
    // dense caller-order arrays get them through ids_map() (rank of every id in use: owned first, then ghosts)
    int id_space = 0;
    bool id_gaps = false;
    DevBuf<int> cmap;
    bool cmap_valid = false;
    // typed boxes (two species): sort digit = cell * nt + species; count[] then holds the per-(cell, species) starts and
    // cstart[] the per-cell ones every untyped consumer reads
    SpeciesTable species{1, {0, 0, 0, 0}};
    int nt = 1;
    // untyped boxes on the brick path: a cell's atoms ordered by quarter along x (kernels.hpp XSubBin), digit = cell * nsub + quarter
    int nsub = 1;
    bool subbins_enabled = std::getenv("EMDEE_NO_SUBBINS") == nullptr && exp_env("EMDEE_NO_BRICK_TABLES") == nullptr;
    // two-species boxes on the brick path (round 5): the typed build takes the same quarters, digit = (cell * 2 + species) * 4 + quarter
    // (EMDEE_TYPED_SUBBINS=0: the (cell, species) order of rounds 3-4, the A/B baseline)
    int tsub = 1;
    bool typed_subbins_enabled = std::getenv("EMDEE_TYPED_SUBBINS") == nullptr || std::atoi(std::getenv("EMDEE_TYPED_SUBBINS")) != 0;
    int digits() const { return nt > 1 ? nt * tsub : nsub; }
    DevBuf<int> bsub;
    bool typed_enabled = std::getenv("EMDEE_NO_TYPED") == nullptr;
    DevBuf<int> cstart;
    DevBuf<unsigned long long> species_tab;
    DevBuf<unsigned short> nbr16;
    DevBuf<int> btab;                     // per-brick tables of the current list (k_brick_tables)
    bool btab_valid = false;
    DevBuf<double> partial, sums;
    DevBuf<unsigned long long> stats;
    Scanner scanner;

    NbSystem() {
        refuse_experiment_switches();
        if (const char *e = std::getenv("EMDEE_PATH")) path = (std::string(e) == "direct") ? PATH_DIRECT : PATH_BRICK;
        if (const char *e = exp_env("EMDEE_BRICK_VARIANT")) {
            variant = std::max(0, std::min(BRICK_VARIANTS - 1, std::atoi(e)));
            variant_forced = true;
        }
        if (const char *e = exp_env("EMDEE_BUILD_ALG")) { force_build1 = std::atoi(e) == 1; if (std::atoi(e) == 2) build_alg_pref = 2; }
        if (const char *e = std::getenv("EMDEE_RUN_AHEAD")) run_ahead = std::max(1, std::min(RUN_AHEAD, std::atoi(e)));
    }

    hipStream_t stream() const { return ctx->stream; }
    AtomView<real> view() const { return AtomView<real>{rec.ptr, te.ptr, rel_grid(rel_now, cell_sorted.ptr)}; }

    // ---- cell-relative records (kernels.hpp RelGrid): fp32 integrators on the tiled path, undivided boxes ----
    // rel_now: what the records of the CURRENT sorted state are; decided anew at every sort (rel_wanted), the gather kernels
    // read one representation and write the other.  EMDEE_F32_ABS=1 keeps absolute fp32 records (the A/B baseline: rounds 1-4).
    bool rel_now = false;
    double rel_lo[3] = {0, 0, 0}, rel_cw[3] = {1, 1, 1};
    int rel_M[3] = {1, 1, 1};
    bool rel_wanted() const {
        return sizeof(real) == 4 && with_vel && !with_mass && !use_tags && !has_ghosts && path == PATH_BRICK && !tbuild_enabled &&
               near_far_scale() <= 0.0 && std::getenv("EMDEE_F32_ABS") == nullptr;
    }
    RelGrid rel_grid(bool on, const int *cells) const {
        RelGrid r{};
        for (int d = 0; d < 3; d++) { r.lo[d] = rel_lo[d]; r.cw[d] = rel_cw[d]; r.M[d] = rel_M[d]; }
        r.on = on ? 1 : 0;
        r.cell = cells;
        return r;
    }
    // the grid the NEXT sorted state's records will be relative to (call after configure_grid)
    RelGrid rel_grid_next(bool on, const int *cells) const {
        RelGrid r{};
        for (int d = 0; d < 3; d++) { r.lo[d] = (double)grid.lo[d]; r.cw[d] = (double)grid.len[d] / (double)grid.M[d]; r.M[d] = grid.M[d]; }
        r.on = on ? 1 : 0;
        r.cell = cells;
        return r;
    }
    void rel_commit(bool on) {
        rel_now = on;
        for (int d = 0; d < 3; d++) { rel_lo[d] = (double)grid.lo[d]; rel_cw[d] = (double)grid.len[d] / (double)grid.M[d]; rel_M[d] = grid.M[d]; }
    }

    struct Timed {
        NbSystem *s;
        int id;
        size_t k = 0;
        Timed(NbSystem *sys, int which) : s(sys), id(which) {
            if (s->profiling) k = s->timers[id].begin(s->stream());
        }
        ~Timed() {
            if (s->profiling) s->timers[id].end(k, s->stream());
        }
    };

    // ---------------------------------------------------------------- geometry
    void set_box(const double lo_[3], const double len_[3], const int per_[3]) {
        bool same = true;
        for (int d = 0; d < 3; d++) same = same && lo[d] == lo_[d] && len[d] == len_[d] && per[d] == (per_[d] ? 1 : 0);
        if (!same) sorted = has_list = plan_valid = false;
        for (int d = 0; d < 3; d++) {
            lo[d] = lo_[d]; len[d] = len_[d]; per[d] = per_[d] ? 1 : 0;
            EMDEE_REQUIRE(len[d] > 0.0 && std::isfinite(len[d]), EMDEE_ERR_INVALID, "box length must be positive");
        }
    }

    void set_model(const emdee_lj_model &m, double skin_) {
        EMDEE_REQUIRE(m.rc2 > 0 && m.rs2 >= 0 && m.rs2 < m.rc2 && std::isfinite(m.inv_delta2), EMDEE_ERR_INVALID,
                      "LennardJonesModel needs 0 <= switch < cutoff (Q10: switch == cutoff gives 1/0)");
        EMDEE_REQUIRE(skin_ >= 0.0, EMDEE_ERR_INVALID, "skin must be >= 0");
        if (m.rc2 != model_d.rc2 || skin_ != skin) has_list = sorted = plan_valid = false;
        model_d = m;
        model = make_model<real>(m);
        skin = skin_;
        rlist = std::sqrt(m.rc2) + skin;
    }

    void configure_grid() {
        GridP<real> g{};
        size_t cells = 1;
        int M[3];
        for (int d = 0; d < 3; d++) {
            if (per[d])
                EMDEE_REQUIRE(rlist <= 0.5 * len[d], EMDEE_ERR_INVALID,
                              "cutoff + skin = %g exceeds half the periodic box length %g (minimum image)", rlist, len[d]);
            M[d] = std::max(1, (int)std::floor(len[d] / rlist));   // cell side >= rlist: 27-cell stencil
            M[d] = std::min(M[d], 1024);
        }
        // sparse boxes: never more cells than ~4 per atom (larger cells stay valid)
        while ((size_t)M[0] * M[1] * M[2] > 4 * (size_t)std::max(n_total, 16) + 64) {
            int big = 0;
            for (int d = 1; d < 3; d++)
                if (M[d] > M[big]) big = d;
            if (M[big] <= 1) break;
            M[big] = (M[big] + 1) / 2;
        }
        for (int d = 0; d < 3; d++) {
            g.lo[d] = (real)lo[d]; g.len[d] = (real)len[d];
            g.plen[d] = per[d] ? (real)len[d] : (real)0;
            g.pinv[d] = per[d] ? (real)(1.0 / len[d]) : (real)0;
            g.per[d] = per[d];
            g.M[d] = M[d];
            cells *= (size_t)M[d];
        }
        g.nd = 1;
        g.one_based = 0;
        grid = g;
        ncell = cells;
    }

    void reserve(int n, bool velocities, bool masses) {
        n_total = n;
        const size_t m = std::max((size_t)n, cap_hint);
        capacity = m;
        pitch = (m + 63) / 64 * 64 + 64;
        with_vel = velocities;
        with_mass = masses;
        rec.ensure(m + 1); rec2.ensure(m + 1);
        if (sizeof(real) == 4) { te.ensure(m + 1); te2.ensure(m + 1); }
        frc.ensure(3 * pitch); en.ensure(pitch); vir.ensure(pitch); xb.ensure(3 * pitch);
        if (velocities) { vel.ensure(3 * pitch); vel2.ensure(3 * pitch); }
        if (masses) { im.ensure(pitch); im2.ensure(pitch); }
        perm.ensure(m + 1); perm2.ensure(m + 1); inv_perm.ensure(m + 1); img.ensure(m + 1); img2.ensure(m + 1);
        cell_of.ensure(m + 1); cell_sorted.ensure(m + 1); cell_sorted2.ensure(m + 1); order.ensure(m + 1); cnt.ensure(m + 1);
        if (use_tags) { tag.ensure(m + 1); tag2.ensure(m + 1); }
        if (flags.ensure(32)) EMDEE_HIP_CHECK(hipMemsetAsync(flags.ptr, 0, 32 * sizeof(int), stream()));
        partial.ensure(3 * RED_MAX_BLOCKS); sums.ensure(8); stats.ensure(4);
    }

    // ---------------------------------------------------------------- binning
    // n_items / tagkey / keep: see resort_edit (items struck out by keep[] take no part; the number that do is count[nbins])
    template <class Src, class Spc>
    void bin(Src src, const int *key, Spc spc, int n_items = -1, const long long *tagkey = nullptr, const unsigned char *keep = nullptr) {
        const int n = n_items >= 0 ? n_items : n_total;
        const int dm = digits();
        const size_t nbins = ncell * (size_t)dm;
        count.ensure(nbins + 2); fill.ensure(nbins + 2);
        Zeros zr;
        zr.add(count.ptr, nbins + 1).add(fill.ptr, nbins);
        if (dm > 1) {
            cstart.ensure(ncell + 2);
            if (n == 0) zr.add(cstart.ptr, ncell + 1);
        }
        zr.run(stream());
        if (n == 0) return;
        hipLaunchKernelGGL((k_cell_assign<real, Src, Spc>), dim3(blocks_for(n, 256)), dim3(256), 0, stream(), n, src, grid,
                           cell_of.ptr, count.ptr, spc, dm, keep);
        scanner.run(count.ptr, nbins + 1, stream());   // count[] becomes the start of every (cell, species / sub-bin) block
        tmp2.ensure(std::max((size_t)n, capacity) + 1);
        hipLaunchKernelGGL(k_cell_scatter_keyed, dim3(blocks_for(n, 256)), dim3(256), 0, stream(), n, cell_of.ptr, count.ptr,
                           fill.ptr, key, tmp2.ptr, tagkey);
        hipLaunchKernelGGL(k_cell_rankfix_keyed, dim3(blocks_for(n, 256)), dim3(256), 0, stream(), n, cell_of.ptr, count.ptr,
                           tmp2.ptr, order.ptr, keep ? count.ptr + nbins : nullptr, tagkey != nullptr ? 1 : 0);
        // (the first slot of every cell, cstart[], is written by the gather kernel that follows)
    }
    // K: how many quarters of a neighbour cell lie beyond r_list whatever the atom's position in its own quarter: the
    // quarters < s + K of the left cell are at least (1 + (K - 1) / 4) cell widths away.  0 for a cell a little wider than
    // r_list; -1 (one more quarter on either side) when the cell is r_list to within the margin that covers the rounding
    // of M t in the box's own precision (t carries ~2 ulp of 1, M t as many ulp of M: 16 M eps is 8e-5 for the fp32 10^7-atom
    // box, whose cells are 8.6e-4 wider than r_list)
    int sub_k() const {
        const double cx = len[0] / std::max(1, grid.M[0]);
        const double margin = 16.0 * grid.M[0] * (sizeof(real) == 4 ? 6.0e-8 : 1.2e-16) + 1e-9;
        return std::max(-1, (int)std::floor(4.0 * (1.0 - rlist * (1.0 + margin) / cx)));
    }
    // sub-bins only where the round-robin two-phase build can use them: the tiled path of an untyped box
    void choose_subbins() {
        nsub = (subbins_enabled && path == PATH_BRICK && nt == 1 && n_total > 0) ? 4 : 1;
        tsub = (subbins_enabled && typed_subbins_enabled && typed_enabled && path == PATH_BRICK && nt == 2 && n_total > 0) ? 4 : 1;
    }
    template <class Src, class Spc>
    void bin_species(Src src, const int *key, Spc spc, int n_items = -1, const long long *tagkey = nullptr, const unsigned char *keep = nullptr) {
        using Sub = XSubBin<real, Src>;
        if (tsub > 1) bin(src, key, SpeciesSub<Spc, Sub>{spc, Sub{src, grid.lo[0], grid.len[0], grid.M[0], grid.per[0], tsub}}, n_items, tagkey, keep);
        else bin(src, key, spc, n_items, tagkey, keep);
    }
    template <class Src>
    void bin_untyped(Src src, const int *key, int n_items = -1, const long long *tagkey = nullptr, const unsigned char *keep = nullptr) {
        if (nsub > 1) bin(src, key, XSubBin<real, Src>{src, grid.lo[0], grid.len[0], grid.M[0], grid.per[0], nsub}, n_items, tagkey, keep);
        else bin(src, key, NoSpecies{}, n_items, tagkey, keep);
    }
    const int *start() const { return digits() > 1 ? cstart.ptr : count.ptr; }   // first slot of every cell
    const int *tstart() const { return count.ptr; }                        // ... of every (cell, species) block (typed boxes)

    // caller-order arrays -> cell-ordered state (+ list)
    // (tags_user, optional: 64-bit ids in caller order, owned atoms and ghosts -- see `tag`)
    void load_user(int n_own, int n_ghost, const real *pos, const real *velocities, const emdee_lj_atom *atoms,
                   const real *inv_mass, const long long *tags_user = nullptr) {
        EMDEE_REQUIRE(n_own >= 0 && n_ghost >= 0, EMDEE_ERR_INVALID, "negative atom count");
        EMDEE_REQUIRE((int64_t)n_own + n_ghost < (int64_t)1 << 31, EMDEE_ERR_INVALID, "too many atoms");
        EMDEE_REQUIRE(n_own + n_ghost == 0 || (pos && atoms), EMDEE_ERR_INVALID, "positions/atoms are NULL");
        Timed t(this, T_REBUILD);
        n_owned = n_own;
        has_ghosts = n_ghost > 0;
        id_space = n_own + n_ghost; id_gaps = false; cmap_valid = false;
        lgv_ids = nullptr;                                   // caller-order array of the previous state
        use_tags = tags_user != nullptr;
        reserve(n_own + n_ghost, velocities != nullptr || with_vel, inv_mass != nullptr);
        detect_uniform_atoms(atoms);
        configure_grid();
        const int n = n_total;
        const bool rel_next = rel_wanted();
        choose_subbins();
        if (nt > 1) bin_species(UserPos<real>{pos}, nullptr, UserSpecies{species, atoms}, -1, use_tags ? tags_user : nullptr);
        else bin_untyped(UserPos<real>{pos}, nullptr, -1, use_tags ? tags_user : nullptr);
        if (n > 0)
            hipLaunchKernelGGL((k_gather_user<real>), dim3(blocks_for(n, 256)), dim3(256), 0, stream(), n, n_owned, pitch,
                               grid, order.ptr, cell_of.ptr, pos, atoms, velocities, inv_mass, rec.ptr, te.ptr, xb.ptr,
                               with_vel ? vel.ptr : nullptr, with_mass ? im.ptr : nullptr, perm.ptr, inv_perm.ptr,
                               cell_sorted.ptr, img.ptr, digits(), use_tags ? tags_user : nullptr, use_tags ? tag.ptr : nullptr,
                               (int)ncell, digits() > 1 ? cstart.ptr : nullptr, count.ptr, rel_grid_next(rel_next, nullptr));
        rel_commit(rel_next);
        // ghosts are never written by the step kernel: both position buffers carry their records (LJAtom fields)
        // from the start; their coordinates are refreshed by every halo unpack
        if (has_ghosts)
            hipLaunchKernelGGL((k_copy_ghost_records<real>), dim3(blocks_for(n, 256)), dim3(256), 0, stream(), n, n_owned,
                               perm.ptr, rec.ptr, rec2.ptr);
        sorted = true;
        build_list();
    }

    // re-bin / re-sort the current state (MD rebuild)
    void resort() {
        EMDEE_REQUIRE(sorted, EMDEE_ERR_STATE, "no state loaded");
        Timed t(this, T_REBUILD);
        const int n = n_total;
        configure_grid();
        choose_subbins();
        const long long *tk = use_tags ? tag.ptr : nullptr;
        const RelGrid rel_in = rel_grid(rel_now, cell_sorted.ptr);   // (the cells the current records are relative to: read before the grid may change)
        const bool rel_next = rel_wanted();
        if (nt > 1) bin_species(RecPos<real>{rec.ptr, rel_in}, perm.ptr, RecSpecies<real>{species, rec.ptr, te.ptr}, -1, tk);
        else bin_untyped(RecPos<real>{rec.ptr, rel_in}, perm.ptr, -1, tk);
        if (n > 0)
            hipLaunchKernelGGL((k_gather_sorted<real, false>), dim3(blocks_for(n, 256)), dim3(256), 0, stream(), n, pitch, grid,
                               order.ptr, cell_of.ptr, rec.ptr, te.ptr, with_vel ? vel.ptr : nullptr,
                               with_mass ? im.ptr : nullptr, perm.ptr, img.ptr, rec2.ptr, te2.ptr, xb.ptr,
                               with_vel ? vel2.ptr : nullptr, with_mass ? im2.ptr : nullptr, perm2.ptr, inv_perm.ptr,
                               cell_sorted2.ptr, img2.ptr, digits(), tk, use_tags ? tag2.ptr : nullptr, (const int *)nullptr,
                               (int *)nullptr, (int)ncell, digits() > 1 ? cstart.ptr : nullptr, count.ptr, rel_in,
                               rel_grid_next(rel_next, nullptr));
        swap_sorted_buffers();
        rel_commit(rel_next);
        build_list();
    }
    void swap_sorted_buffers() {
        rec.swap(rec2); te.swap(te2); perm.swap(perm2); img.swap(img2); cell_sorted.swap(cell_sorted2);
        if (with_vel) vel.swap(vel2);
        if (with_mass) im.swap(im2);
        if (use_tags) tag.swap(tag2);
    }

    // ---------------------------------------------------------------- a decomposed domain's rebuild in its own order (dd.hpp)
    // The state is re-sorted from its OWN records -- no caller-order copy, no assembly of new caller arrays, no load:
    //  * keep[q] == 0 strikes slot q out (atoms that left for a neighbour, the old ghosts, the unused rows of padded messages);
    //  * the arrivals and the new ghosts have been written behind the old state, slots [n_total, n_items), by the caller --
    //    records (both LJAtom fields), velocity planes, tags;
    //  * the OLD slot q is an atom's id from now on: ids >= own_limit are ghosts (the ids in use have gaps: ids_map()).
    // How many atoms that leaves is a device word (*n_live_dev) until the build's read-back, which carries the caller's words
    // (`extra`: counts and overflow words of the messages) along: ONE blocking read-back for the whole rebuild.  The kernels in
    // between run over n_items.  The old state stays intact in the spare buffers until commit_edit(): when a message of the
    // rebuild turns out to have overflowed, rollback_edit() puts it back and the caller redoes the rebuild with counts.
    struct EditWords {
        const int *dev = nullptr;
        int n = 0;
        int32_t *host = nullptr;
    };
    bool in_edit = false, edit_abort = false;
    EditWords edit_extra{};
    int edit_n_plan = 0;
    struct EditSaved { int n_total, n_owned, id_space; bool id_gaps, has_ghosts, has_list; } edit_saved{};
    bool edit_fits(int n_items) const { return sorted && use_tags && with_vel && !with_mass && (size_t)n_items <= capacity; }
    bool edit_extra_read = false;
    bool resort_edit(int n_items, const unsigned char *keep, int own_limit, bool ghosts, int *n_live_dev, EditWords extra) {
        EMDEE_REQUIRE(edit_fits(n_items), EMDEE_ERR_STATE, "resort_edit: the state does not hold %d slots", n_items);
        Timed t(this, T_REBUILD);
        edit_saved = EditSaved{n_total, n_owned, id_space, id_gaps, has_ghosts, has_list};
        configure_grid();
        choose_subbins();
        EMDEE_REQUIRE(!rel_now, EMDEE_ERR_STATE, "resort_edit: cell-relative records (decomposed domains keep absolute ones)");
        if (nt > 1) bin_species(RecPos<real>{rec.ptr, RelGrid{}}, nullptr, RecSpecies<real>{species, rec.ptr, te.ptr}, n_items, tag.ptr, keep);
        else bin_untyped(RecPos<real>{rec.ptr, RelGrid{}}, nullptr, n_items, tag.ptr, keep);
        const size_t nbins = ncell * (size_t)digits();
        hipLaunchKernelGGL((k_gather_sorted<real, true>), dim3(blocks_for(n_items, 256)), dim3(256), 0, stream(), n_items, pitch, grid,
                           order.ptr, cell_of.ptr, rec.ptr, te.ptr, vel.ptr, (const real *)nullptr, perm.ptr, img.ptr, rec2.ptr, te2.ptr,
                           xb.ptr, vel2.ptr, (real *)nullptr, perm2.ptr, inv_perm.ptr, cell_sorted2.ptr, img2.ptr, digits(), tag.ptr,
                           tag2.ptr, count.ptr + nbins, n_live_dev, (int)ncell, digits() > 1 ? cstart.ptr : nullptr, count.ptr);
        swap_sorted_buffers();
        edit_n_plan = n_total;
        n_total = n_items;                                  // an upper bound until commit_edit()
        n_owned = own_limit;
        has_ghosts = ghosts;
        id_space = n_items; id_gaps = true; cmap_valid = false;
        in_edit = true; edit_abort = false; edit_extra = extra; edit_extra_read = false;
        struct Leave { bool &f; ~Leave() { f = false; } } leave{in_edit};
        // (an engine on the direct kernels, or without atoms so far: they count atoms on the host -- the re-sorted state is
        // complete, the caller reads the counts and loads the engine from it)
        if (path == PATH_BRICK && plan_valid && edit_saved.n_total > 0) build_list();
        else edit_abort = true;
        if (edit_abort && !edit_extra_read && extra.n > 0) read_back_words(ctx, stream(), extra.dev, extra.n, extra.host);
        return !edit_abort;
    }
    void commit_edit(int n_live) {
        EMDEE_REQUIRE(n_live >= 0 && n_live <= n_total, EMDEE_ERR_STATE, "resort_edit: %d atoms in %d slots", n_live, n_total);
        n_total = n_live;
        // ghosts are never written by the step kernel: both position buffers carry their records from the start (only now:
        // until here the spare buffer held the state to roll back to)
        if (has_ghosts && n_total > 0)
            hipLaunchKernelGGL((k_copy_ghost_records<real>), dim3(blocks_for(n_total, 256)), dim3(256), 0, stream(), n_total, n_owned,
                               perm.ptr, rec.ptr, rec2.ptr);
    }
    void rollback_edit() {
        swap_sorted_buffers();
        n_total = edit_saved.n_total; n_owned = edit_saved.n_owned; id_space = edit_saved.id_space;
        id_gaps = edit_saved.id_gaps; has_ghosts = edit_saved.has_ghosts;
        cmap_valid = false;
        has_list = false; plan_valid = false; btab_valid = false;   // (inv_perm, xb, the cell tables and the list are the discarded sort's)
    }
    // rank of every id in use among the ids in use (NULL: the ids have no gaps)
    const int *ids_map() {
        if (!id_gaps) return nullptr;
        if (!cmap_valid) {
            cmap.ensure((size_t)id_space + 2);
            EMDEE_HIP_CHECK(hipMemsetAsync(cmap.ptr, 0, ((size_t)id_space + 1) * sizeof(int), stream()));
            if (n_total > 0) hipLaunchKernelGGL(k_mark_live, dim3(blocks_for(n_total, 256)), dim3(256), 0, stream(), n_total, perm.ptr, cmap.ptr);
            scanner.run(cmap.ptr, (size_t)id_space + 1, stream());
            cmap_valid = true;
        }
        return cmap.ptr;
    }

    // ---------------------------------------------------------------- brick plumbing
    BrickArgs<real> brick_args(int phase = 0) {
        BrickArgs<real> a{};
        a.n = n_total; a.n_owned = n_owned; a.any_ghosts = has_ghosts ? 1 : 0;
        a.rec = rec.ptr; a.te = te.ptr; a.perm = perm.ptr; a.start = start();
        a.g = grid; a.bg = bgrid; a.tile_cap = tile_cap; a.own_cap = own_cap;
        a.nbr = nbr16.ptr; a.stride = stride; a.cnt = cnt.ptr; a.flags = flags.ptr;
        a.rlist2 = (real)(rlist * rlist); a.margin = build_margin; a.model = model; a.pitch = pitch;
        {
            const double k = near_far_scale();
            a.nf_scale = k > 0.0 ? (float)k : 1.f;
            a.nf_scale2 = a.nf_scale * a.nf_scale;     // (the square of the fp32 scale the tile really gets)
        }
        if (const char *dbg = exp_env("EMDEE_DEBUG_RC2_SCALE")) a.model.rc2 = (real)(std::atof(dbg) * (double)model.rc2);   // ablation only (EXPERIMENTS builds)
        a.frc = frc.ptr; a.en = en.ptr; a.vir = vir.ptr; a.stats = stats.ptr;
        a.phase = phase;   // only force launches are phased; build and stats always cover every brick
        a.vel = vel.ptr; a.vel_next = vel2.ptr; a.xb = xb.ptr; a.inv_mass = with_mass ? im.ptr : nullptr; a.rec_next = rec2.ptr;
        a.kick_c = (real)step_c; a.dt = (real)step_dt;
        a.uni_sigma2 = (real)uni_sigma2; a.uni_e4 = (real)uni_e4;
        a.uni = make_uni<real>(model, (real)uni_sigma, (real)uni_e4);
        a.idx_shift = idx_shift;
        a.tstart = tstart(); a.tdig = nt > 1 ? tsub : 1;
        a.nsub = nsub; a.fstart = tstart(); a.bsub = nsub > 1 ? bsub.ptr : nullptr;
        a.sub_k = sub_k();
        for (int q = 0; q < 4; q++) { a.tsig2[q] = (real)0; a.te4[q] = (real)0; }
        if (nt == 2) {
            // sigma_ij^2 and 4 eps_ij of the four species pairs, with the operations the general-species pair loop uses
            for (int i = 0; i < 2; i++)
                for (int j = 0; j < 2; j++) {
                    float hi, ti, hj, tj;
                    species_atom(i, hi, ti); species_atom(j, hj, tj);
                    const real sg = (real)hi + (real)hj;
                    a.tsig2[i * 2 + j] = sg * sg;
                    a.te4[i * 2 + j] = (real)ti * (real)tj;
                }
        }
        a.rel = rel_now ? 1 : 0;
        for (int d = 0; d < 3; d++) { a.rcw[d] = rel_cw[d]; a.rlo[d] = rel_lo[d]; }
        a.far_skip = far_skip_active() ? 1 : 0;
        a.far_word = flags.ptr + 16;
        a.thr2_near = (real)(0.25 * near_delta() * near_delta());
        a.refmath = (sizeof(real) == 4 && refmath && ref_pos != nullptr) ? 1 : 0;
        a.user_pos = ref_pos;
        a.thr2 = (real)(0.25 * skin * skin);
        a.trigger = step_trigger ? step_trigger : flags.ptr + 1;
        a.guard = step_guard;
        a.btab = btab_valid ? btab.ptr : nullptr;
        a.user_f = out_f; a.user_e = out_e; a.user_w = out_w;
        a.noise = lgv_on ? noise.ptr : nullptr; a.lgv_c1 = lgv_on ? (real)lgv_c1 : (real)1;
        return a;
    }

    void species_atom(int k, float &hs, float &te) const {
        const unsigned lo = (unsigned)(species.key[k] & 0xffffffffull), hi = (unsigned)(species.key[k] >> 32);
        memcpy(&hs, &lo, 4); memcpy(&te, &hi, 4);
    }
    // typed boxes (typed.hpp): variants 0 and 7 (4 lanes per atom, 512 / 1024 threads) carry the two-species kernels; masks other
    // than FORCES run the all-outputs kernel
    bool typed_active = false;
    template <class V>
    static constexpr bool typed_variant() {
        return std::is_same<V, BrickVariant<0>>::value || std::is_same<V, BrickVariant<7>>::value || std::is_same<V, BrickVariant<9>>::value;
    }
    template <class V>
    static constexpr int typed_min_stride() { return (typed_prefetch_blocks(V::G, V::THREADS, V::Shape::NOC) + 1) * EPL * V::G; }
    template <class V, int MODE, int BM>
    void launch_typed_kernel() {
        if constexpr (typed_variant<V>()) {
            constexpr int M = (MODE == BRICK_STATS) ? 0 : ((MODE == BRICK_STEP || BM == 1) ? 1 : 7);
            auto kernel = k_typed<real, typename V::Shape, V::THREADS, V::G, MODE, M>;
            const size_t lds = typed_force_lds_bytes<real, typename V::Shape, V::THREADS>(own_cap);
            allow_big_lds(kernel, lds);
            const int phase = (MODE == BRICK_FORCE || MODE == BRICK_STEP) ? force_phase : 0;
            const int blocks = (phase == 1 ? bgrid.ib_per_xcd : phase == 2 ? bgrid.bb_per_xcd : bgrid.per_xcd) * NXCD;
            if (blocks == 0) return;
            hipLaunchKernelGGL(kernel, dim3(blocks), dim3(V::THREADS), lds, stream(), brick_args(phase));
        }
    }

    template <class V, int MODE, int BM>
    void launch_brick_kernel() {
        if (typed_active) { launch_typed_kernel<V, MODE, BM>(); return; }
        // single-species fast path for the kernels of the MD loop (default variant only)
        if constexpr (std::is_same<V, BrickVariant<0>>::value && (MODE == BRICK_STEP || (MODE == BRICK_FORCE && (BM == 1 || BM == 7)))) {
            // (the fp64 variant keeps coordinate planes only in LDS and needs the tile to fit their fixed pitch)
            if (uniform_atoms && tile_cap <= SOA_SLOTS && idx_shift == PLANE_SHIFT && !(MODE == BRICK_FORCE && sizeof(real) == 4 && refmath && ref_pos != nullptr)) {
                launch_brick_kernel_impl<V, MODE, BM, true>();
                return;
            }
        }
        launch_brick_kernel_impl<V, MODE, BM, false>();
    }

    template <class V, int MODE, int BM, bool UNI>
    void launch_brick_kernel_impl() {
        auto kernel = k_brick<real, typename V::Shape, V::THREADS, V::G, MODE, BM, UNI>;
        const size_t lds = (UNI && MODE != BRICK_STATS) ? brick_force_lds_bytes_soa<real, typename V::Shape, V::THREADS>(own_cap)
                                                        : lds_bytes;
        allow_big_lds(kernel, lds);
        const int phase = (MODE == BRICK_FORCE || MODE == BRICK_STEP) ? force_phase : 0;
        const int blocks = (phase == 1 ? bgrid.ib_per_xcd : phase == 2 ? bgrid.bb_per_xcd : bgrid.per_xcd) * NXCD;
        if (blocks == 0) return;
        hipLaunchKernelGGL(kernel, dim3(blocks), dim3(V::THREADS), lds, stream(), brick_args(phase));
    }

    template <class V>
    void launch_brick_force(int bitmask) {
        if constexpr (std::is_same<V, BrickVariant<0>>::value) {
            switch (bitmask) {
                case 1: launch_brick_kernel<V, BRICK_FORCE, 1>(); return;
                case 2: launch_brick_kernel<V, BRICK_FORCE, 2>(); return;
                case 3: launch_brick_kernel<V, BRICK_FORCE, 3>(); return;
                case 4: launch_brick_kernel<V, BRICK_FORCE, 4>(); return;
                case 5: launch_brick_kernel<V, BRICK_FORCE, 5>(); return;
                case 6: launch_brick_kernel<V, BRICK_FORCE, 6>(); return;
                default: launch_brick_kernel<V, BRICK_FORCE, 7>(); return;
            }
        }
        // tuning variants carry only the two masks the MD loop uses
        if (bitmask == 1) launch_brick_kernel<V, BRICK_FORCE, 1>();
        else launch_brick_kernel<V, BRICK_FORCE, 7>();
    }

    // brick decomposition + LDS tile capacity for the current cell populations; false if a tile cannot fit in LDS (then
    // the direct kernels are used).  Three parts: the brick grid (geometry only), the population maxima (one small kernel,
    // results in flags[6..8]), and the sizes derived from them on the host.
    void plan_geometry() {
        with_brick_variant(variant, [&](auto v) {
            using V = decltype(v);
            using S = typename V::Shape;
            bgrid.nb[0] = (grid.M[0] + S::BX - 1) / S::BX;
            bgrid.nb[1] = (grid.M[1] + S::BY - 1) / S::BY;
            bgrid.nb[2] = (grid.M[2] + S::BZ - 1) / S::BZ;
            bgrid.nbricks = bgrid.nb[0] * bgrid.nb[1] * bgrid.nb[2];
            bgrid.per_xcd = (bgrid.nbricks + NXCD - 1) / NXCD;
            // interior bricks: along an open dimension, those whose tile (brick +- 1 cell) stays clear of
            // the outermost cell layers, where the ghosts of a decomposed run live
            const int B[3] = {S::BX, S::BY, S::BZ};
            bgrid.ib_count = 1;
            for (int d = 0; d < 3; d++) {
                int lo_b = 0, n_b = bgrid.nb[d];
                if (!grid.per[d]) {
                    lo_b = 1;
                    int hi_b = 1;
                    while (hi_b < bgrid.nb[d] && (hi_b + 1) * B[d] < grid.M[d] - 1) hi_b++;
                    n_b = std::max(0, hi_b - lo_b);
                }
                bgrid.ib_lo[d] = lo_b; bgrid.ib_n[d] = n_b;
                bgrid.ib_count *= n_b;
            }
            bgrid.ib_per_xcd = (bgrid.ib_count + NXCD - 1) / NXCD;
            bgrid.bb_z = (bgrid.nb[2] - bgrid.ib_n[2]) * bgrid.nb[0] * bgrid.nb[1];
            bgrid.bb_y = bgrid.ib_n[2] * (bgrid.nb[1] - bgrid.ib_n[1]) * bgrid.nb[0];
            bgrid.bb_count = bgrid.nbricks - bgrid.ib_count;
            bgrid.bb_per_xcd = (bgrid.bb_count + NXCD - 1) / NXCD;
            row_block = EPL * V::G;
        });
    }
    // flags[6] = largest tile, flags[7] = most own atoms of a brick, flags[8] = most atoms in three consecutive cells of a tile row
    // (with_build_words: the build's own words flags[0..5) are cleared by the same launch -- the first attempt of build_list)
    void launch_tile_max(bool with_build_words = false) {
        with_brick_variant(variant, [&](auto v) {
            using S = typename decltype(v)::Shape;
            if (with_build_words) Zeros().add(flags.ptr, 9).run(stream());
            else Zeros().add(flags.ptr + 6, 3).run(stream());
            hipLaunchKernelGGL((k_brick_tile_max<S>), dim3(blocks_for(bgrid.nbricks, 256)), dim3(256), 0, stream(), bgrid,
                               grid.M[0], grid.M[1], grid.M[2], grid.per[0], grid.per[1], grid.per[2], start(),
                               flags.ptr + 6);
        });
    }
    // A plan is kept from one rebuild to the next (the populations barely change): capacities carry a few percent of
    // headroom, and the maxima of the NEW populations are checked after the build, together with its overflow words.
    static int with_headroom(int v) { return (v + v / 32 + 8 + 15) / 16 * 16; }
    bool plan_sizes(int max_tile, int max_own, int max_span3) {
        bool ok = true;
        with_brick_variant(variant, [&](auto v) {
            using V = decltype(v);
            using S = typename V::Shape;
            tile_cap = std::max(64, with_headroom(max_tile + 1));   // + 1: the sentinel record
            // (the single-species kernels keep coordinate planes of SOA_SLOTS records: do not let the headroom alone push a tile past them)
            if (max_tile + 1 <= SOA_SLOTS) tile_cap = std::min(tile_cap, SOA_SLOTS);
            own_cap = std::max(64, with_headroom(max_own));
            {   // ... nor cost a workgroup per CU in the build or force kernels: give headroom back 16 records at a time
                const int exact = std::max(64, (max_tile + 1 + 15) / 16 * 16), st = stride > 0 ? stride : 128;
                // (a CU holds three workgroups only up to ~50,000 B each, not 160 KB / 3: measured in round 2 by padding the launch)
                constexpr size_t USABLE = 150000;
                auto per_cu = [&](int tc) {
                    size_t b = brick_build_lds_bytes<S, V::THREADS>(tc, own_cap, st, V::GB, nsub);
#ifdef EMDEE_EXPERIMENTS
                    if (tbuild_active<V>()) b = brick_tbuild_lds_bytes<S, V::THREADS, TB_NPAIR>(tc, own_cap, st);
#endif
                    const size_t
                                 f = brick_force_lds_bytes<real, S, V::THREADS>(tc, own_cap);
                    return (int)(USABLE / std::max<size_t>(b, 1)) * 16 + (int)(USABLE / std::max<size_t>(f, 1));
                };
                while (tile_cap > exact && per_cu(tile_cap) < per_cu(exact)) tile_cap -= 16;
            }
            plan_span3 = max_span3 + max_span3 / 16 + 2;
            if (std::getenv("EMDEE_DEBUG_PLAN"))
                std::fprintf(stderr, "emdee plan: bricks %d x %d x %d, tile_cap %d (max %d), own_cap %d (max %d), max 3-cell span %d, x sub-bins %d K %d\n", bgrid.nb[0],
                             bgrid.nb[1], bgrid.nb[2], tile_cap, max_tile, own_cap, max_own, max_span3, nsub, sub_k());
            build_alg = (!force_build1 && !field16_blocked && (V::GB == 8 || V::GB == 16) && plan_span3 <= BUILD2_FIELD * V::GB) ? build_alg_pref : 1;
            // crowded tile rows (long cutoffs): the same build with one 32-bit hit field per row
            if (build_alg == 1 && !force_build1 && build_alg_pref == 3 && (V::GB == 8 || V::GB == 16) && plan_span3 <= 32 * V::GB) build_alg = 5;
            if (build_alg == 1) plan_span3 = 1 << 30;               // the ballot build has no limit
            else plan_span3 = (build_alg == 5 ? 32 : BUILD2_FIELD) * V::GB;   // what the chosen build can take
            if (tbuild_active<V>()) plan_span3 = 1 << 30;           // the transposed build reports a crowded cell itself (flags[3])
            lds_bytes = brick_force_lds_bytes<real, S, V::THREADS>(tile_cap, own_cap);
            ok = lds_bytes <= LDS_LIMIT && tile_cap < 65536;
            // fp32 pre-test of the build kernel (fp64 boxes): brick-relative coordinates are below
            // cmax, so each is off by <= cmax 2^-24 after rounding; with |d| <= r_list per component the
            // error of d^2 is below 4 sqrt(3) r_list cmax 2^-24 + 4 r_list^2 2^-23.  Band = 4 x that bound.
            build_margin = 0.f;
            if (sizeof(real) == 8) {
                double cmax = 0.0;
                const int B[3] = {S::BX, S::BY, S::BZ};
                for (int d = 0; d < 3; d++) cmax = std::max(cmax, (B[d] + 2) * len[d] / grid.M[d]);
                const double bound = 4.0 * 1.7320508 * rlist * cmax * std::ldexp(1.0, -24) + 4.0 * rlist * rlist * std::ldexp(1.0, -23);
                build_margin = (float)(4.0 * bound);
            }
        });
        return ok;
    }
    int plan_span3 = 0;                   // most atoms in three consecutive cells of a tile row the chosen build kernel can take
    // do the maxima of the current populations (flags[6..8], read back) fit the plan in use?
    bool plan_holds(const int32_t *maxima) const {
        return maxima[0] + 1 <= tile_cap && maxima[1] <= own_cap && maxima[2] <= plan_span3;
    }
    // first plan of a state (or one that no longer holds): blocking read-back of the maxima
    bool plan_bricks() {
        plan_geometry();
        launch_tile_max();
        read_back_words(ctx, stream(), flags.ptr + 6, 3, ctx->host_flags + 6);
        for (int k = 0; k < 3; k++) plan_maxima[k] = ctx->host_flags[6 + k];
        return plan_sizes(plan_maxima[0], plan_maxima[1], plan_maxima[2]);
    }
    int plan_maxima[3] = {0, 0, 0};

    // ---------------------------------------------------------------- neighbour list
    bool brick_active = false;

    // near/far rows (EMDEE_BUILD_NEARFAR=1; measured in round 3 and left OFF: the force launch gains 6 %, 1.280 -> 1.199 ms,
    // the build loses 0.6 ms, 2.92 -> 3.52, because nine rows x two classes make 18 short emission loops that each run as
    // long as the busiest lane of the wavefront -- 585 vs 584 steps/s; profiles/README.md):
    // r_near = r_c + delta (EMDEE_NEAR_DELTA, default 0.04 length units); returns the scale k of the build tile with
    // k^2 (r_list^2 - r_near^2) = 2, or 0 when switched off or the skin is too thin
    double near_delta() const {
        double delta = 0.04;
        if (const char *e = exp_env("EMDEE_NEAR_DELTA")) delta = std::atof(e);
        return delta;
    }
    // the far class skipped outright while no atom has moved delta / 2 (BrickArgs::far_skip): untyped boxes without ghosts on
    // the near/far build, rows short enough for the two counts to share cnt[p]
    bool far_skip_active() const {
        const char *on = exp_env("EMDEE_FAR_SKIP");
        return on != nullptr && std::atoi(on) != 0 && nearfar_built && brick_active && !typed_active && !has_ghosts && stride < 256;
    }
    bool nearfar_built = false;           // the list in use was written by the near/far build (ALG 23)
    double near_far_scale() const {
        const char *on = exp_env("EMDEE_BUILD_NEARFAR");
        if (on == nullptr || std::atoi(on) == 0) return 0.0;
        const double delta = near_delta();
        const double rc = std::sqrt((double)model_d.rc2), rn = rc + delta;
        if (!(delta >= 0.0) || rn >= rlist - 0.05 * skin) return 0.0;
        return std::sqrt(2.0 / (rlist * rlist - rn * rn));
    }

    bool build_fits_lds() {
        bool ok = true;
        with_brick_variant(variant, [&](auto v) {
            using V = decltype(v);
            ok = brick_build_lds_bytes<typename V::Shape, V::THREADS>(tile_cap, own_cap, stride, V::GB, nsub) <= LDS_LIMIT;
#ifdef EMDEE_EXPERIMENTS
            if (tbuild_active<V>()) ok = brick_tbuild_lds_bytes<typename V::Shape, V::THREADS, TB_NPAIR>(tile_cap, own_cap, stride) <= LDS_LIMIT;
#endif
            if (typed_active) ok = typed_build_lds_bytes<typename V::Shape, V::THREADS>(tile_cap, own_cap, stride, V::GB, V::G) <= LDS_LIMIT;
        });
        return ok;
    }

    // full planning with a blocking read-back: variant, capacities, build kernel, stride rounding, index encoding
    bool plan_valid = false;
    int plan_M[3] = {0, 0, 0}, plan_n = 0;
    void make_plan() {
        const int n = n_total;
        if (!variant_forced) variant = 0;
        brick_active = (path == PATH_BRICK) && n > 0 && plan_bricks();
        // A tile too large for two workgroups of the default variant per CU (long cutoffs, dense boxes: rc = 3.5
        // sigma needs 135 KB) would leave 2 waves per SIMD: take the same bricks with 1024-thread workgroups
        // (measured on the rc = 3.5 mixture: 116 -> 154 steps/s).
        if (brick_active && !variant_forced && lds_bytes > LDS_LIMIT / 2) {
            variant = 8;
            if (!plan_bricks()) { variant = 0; plan_bricks(); }
        }
        if (brick_active) {
            stride = (stride + row_block - 1) / row_block * row_block;   // whole lane-major blocks
            with_brick_variant(variant, [&](auto v) { stride = std::max(stride, brick_min_stride(decltype(v)::G, decltype(v)::THREADS)); });
        }
        // the build kernel's LDS (fp32 tile + tables + one row buffer per lane group) must fit as well: very dense or
        // very inhomogeneous boxes with a long cutoff fall back to the direct (global-gather) kernels
        if (brick_active && !build_fits_lds()) brick_active = false;
        idx_shift = (brick_active && variant == 0 && uniform_atoms && tile_cap <= SOA_SLOTS && !exp_env("EMDEE_NO_PREMUL")) ? PLANE_SHIFT : 0;
        // two species: the typed kernels (typed.hpp), if the tile fits their coordinate planes, no three cells of a tile row hold
        // more atoms of one species than the 16-bit hit fields of their build take, and both kernels fit LDS
        typed_active = false;
        if (brick_active && nt == 2 && !typed_blocked && !has_excl) {
            // (where the general-species kernels take 1024 threads with 8 lanes per atom -- long cutoffs -- the typed ones take
            // 1024 threads with 4: rows are two block-aligned segments, and blocks of 32 entries pad them half as much as blocks of 64)
            // Measured (profiles/README.md, round 3): at rc = 3.5 sigma 190.7 -> 218.0 steps/s in fp64 and 233.8 -> 324.2 in fp32; at
            // rc = 2.5 (variant 0, rows of ~37 entries per species) the 18 short candidate rows cost the build more than the
            // pair loop gains, 453.9 -> 419.1: short-row boxes keep the general-species kernels (EMDEE_TYPED_ALL=1 overrides)
            // Long rows (the general kernels chose 1024-thread workgroups): first the small bricks of variant 9, two 512-thread
            // workgroups per CU -- if its tile and both kernels fit half a CU's LDS --, then variant 7 (EMDEE_TYPED_BRICKS=7: the
            // A/B baseline).  Short rows: variant 0, on request only.
            const int keep = variant;
            int cands[2] = {-1, -1};
            if (variant == 8 && !variant_forced) {
                const char *tb = std::getenv("EMDEE_TYPED_BRICKS");
                const bool only7 = tb != nullptr && std::atoi(tb) == 7;
                cands[0] = only7 ? 7 : 9;
                cands[1] = only7 ? -1 : 7;
            } else if (variant == 7 || variant == 9 || (variant == 0 && std::getenv("EMDEE_TYPED_ALL") != nullptr)) {
                cands[0] = variant;
            }
            const int keep_maxima[3] = {plan_maxima[0], plan_maxima[1], plan_maxima[2]};
            for (int ci = 0; ci < 2 && !typed_active; ci++) {
                const int cand = cands[ci];
                if (cand < 0) continue;
                variant = cand;
                bool same_shape = false;
                with_brick_variant(keep, [&](auto kv) {
                    with_brick_variant(cand, [&](auto cv) { same_shape = std::is_same<typename decltype(kv)::Shape, typename decltype(cv)::Shape>::value; });
                });
                if (same_shape) {
                    for (int k = 0; k < 3; k++) plan_maxima[k] = keep_maxima[k];   // (a candidate of another shape left its own)
                    plan_geometry();
                    plan_sizes(keep_maxima[0], keep_maxima[1], keep_maxima[2]);
                }
                else plan_bricks();                                  // another brick shape: its own population maxima (a read-back; plans are rare)
                with_brick_variant(variant, [&](auto v) {
                    using V = decltype(v);
                    if constexpr (typed_variant<V>()) {
                        using S = typename V::Shape;
                        // (the planes hold typed_slots records: do not let the headroom alone push a tile past them)
                        if (plan_maxima[0] + 1 <= typed_slots<S, V::THREADS>()) tile_cap = std::min(tile_cap, typed_slots<S, V::THREADS>());
                        EMDEE_HIP_CHECK(hipMemsetAsync(flags.ptr + 8, 0, sizeof(int), stream()));
                        hipLaunchKernelGGL((k_typed_span_max<S>), dim3(blocks_for(bgrid.nbricks, 256)), dim3(256), 0, stream(), bgrid,
                                           grid.M[0], grid.M[1], grid.M[2], grid.per[0], grid.per[1], grid.per[2], tstart(), flags.ptr + 8, tsub);
                        EMDEE_HIP_CHECK(hipMemcpyAsync(ctx->host_flags + 8, flags.ptr + 8, sizeof(int), hipMemcpyDeviceToHost, stream()));
                        EMDEE_HIP_CHECK(hipStreamSynchronize(stream()));
                        const int span = ctx->host_flags[8];
                        int st = stride;
                        if (!typed_stride) {
                            st = std::max(stride, typed_min_stride<V>()) + row_block;   // + the padding between the two segments
                            st = (st + row_block - 1) / row_block * row_block;
                        }
                        const size_t lds_f = typed_force_lds_bytes<real, S, V::THREADS>(own_cap),
                                     lds_b = typed_build_lds_bytes<S, V::THREADS>(tile_cap, own_cap, st, V::GB, V::G);
                        // (variant 9 exists to put two workgroups on a CU: it is taken only where both kernels leave room for that)
                        const size_t room = cand == 9 ? LDS_LIMIT / 2 - 128 : LDS_LIMIT;   // (- the kernels' static LDS)
                        const bool ok = tile_cap <= typed_slots<S, V::THREADS>() && span <= 16 * V::GB && lds_f <= room && lds_b <= room;
                        if (std::getenv("EMDEE_DEBUG_PLAN"))
                            std::fprintf(stderr, "emdee plan: two species, variant %d: tile_cap %d (max %d), own_cap %d, longest 3-cell span of one species %d, stride %d, LDS force %zu build %zu: typed kernels %s\n",
                                         cand, tile_cap, plan_maxima[0], own_cap, span, st, lds_f, lds_b, ok ? "on" : "off");
                        if (ok) {
                            typed_active = true;
                            stride = st;
                            typed_stride = true;
                            idx_shift = PLANE_SHIFT;
                        }
                    }
                });
            }
            if (!typed_active && variant != keep) {
                variant = keep;
                plan_bricks();
            }
        }
        if (!typed_active) typed_stride = false;
        plan_valid = brick_active;
        plan_uniform = uniform_atoms;
        plan_nt = nt;
        plan_n = in_edit ? edit_n_plan : n;
        for (int d = 0; d < 3; d++) plan_M[d] = grid.M[d];
    }
    bool plan_uniform = false;
    int plan_nt = 1;
    bool typed_blocked = false;           // this state's rows outgrew the typed build (until the next load)
    bool typed_stride = false;            // the stride already includes the typed rows' segment padding
    bool maxima_from_tables = exp_env("EMDEE_PLAN_MAXIMA") != nullptr && std::string(exp_env("EMDEE_PLAN_MAXIMA")) == "tables";

    void build_list() {
        const int n = n_total;
        if (stride == 0) {
            double vol = len[0] * len[1] * len[2];
            double expect = n > 0 ? (4.0 / 3.0) * M_PI * rlist * rlist * rlist * (double)n / vol : 0.0;
            // first guess: 15 % above the mean row (a jittered lattice at rho* = 0.8, r_list = 2.8 has 73.6 +- 4, longest row 85; the
            // melt 91): rounded up to whole lane-major blocks below, 96 entries there.  A longer row grows the stride and
            // builds again.  (128 instead of 96 costs 2 % of the step: 33 % more bytes flushed per build, rows 256 B apart)
            stride = (int)((expect * 1.15 + 8.0) / 16.0 + 1.0) * 16;
            if (const char *e = exp_env("EMDEE_STRIDE")) stride = std::max(16, std::atoi(e));   // tuning: first guess of the row stride
            typed_stride = false;
        }
        btab_valid = false;
        // The plan of the previous build of this state (variant, capacities, build kernel) is kept when the cell grid is the
        // same: the populations barely change between rebuilds, the capacities carry headroom, and the maxima of the new
        // populations come back with the build's overflow words -- ONE blocking read-back per rebuild instead of two.
        // (inside resort_edit n is an upper bound: the plan is compared with the number of atoms before the edit)
        const int np = in_edit ? edit_n_plan : n;
        bool kept = plan_valid && !exp_env("EMDEE_PLAN_SYNC") && !exp_env("EMDEE_NO_BRICK_TABLES") && path == PATH_BRICK && n > 0 && plan_M[0] == grid.M[0] &&
                    plan_M[1] == grid.M[1] && plan_M[2] == grid.M[2] && plan_n <= np + np / 8 && np <= plan_n + plan_n / 8 &&
                    plan_uniform == uniform_atoms && plan_nt == nt;
        if (kept) {
            plan_geometry();
            // (taking the maxima from k_brick_tables instead, which has every brick's tables in LDS anyway, was measured and
            // lost: one workgroup per brick means one look at -- or atomic on -- three hot words per brick, 2.84 -> 2.95 ms per
            // rebuild even with a device-scope look before the atomic, 4.50 ms without; k_brick_tile_max reduces 64 bricks per
            // wavefront first.  EMDEE_PLAN_MAXIMA=tables switches it on)
            if (maxima_from_tables) Zeros().add(flags.ptr, 9).run(stream());
            else launch_tile_max(true);
            brick_active = true;
        } else {
            make_plan();
        }
        // (a state with a capacity hint -- decomposed domains -- sizes the list for it: no reallocation while atoms come and go)
        const size_t rows = std::max<size_t>((size_t)std::max(n, 1), cap_hint ? capacity : 0);
        for (int attempt = 0; attempt < 6; attempt++) {
            EMDEE_REQUIRE((double)n * stride < 1.7e10, EMDEE_ERR_OVERFLOW, "neighbour list would exceed 64 GiB");
            if (in_edit && !brick_active) { edit_abort = true; return; }   // (the direct kernels count atoms on the host: the caller reloads)
            bool launched4 = false;
            nearfar_built = false;
            if (!(kept && attempt == 0)) Zeros().add(flags.ptr, 5).run(stream());   // (a kept plan cleared them with the maxima)
            if (exp_env("EMDEE_FAR_SKIP")) Zeros().add(flags.ptr + 16, 1).run(stream());   // (experiment: the near word starts afresh)
            if (brick_active) {
                nbr16.ensure(rows * stride);
                with_brick_variant(variant, [&](auto v) {
                    using V = decltype(v);
                    if constexpr (typed_variant<V>()) {
                        if (typed_active) {   // two species: species-major tables and the two-segment build (typed.hpp)
                            constexpr int TT = 256;
                            using BT = TypedTables<typename V::Shape, TT>;
                            static_assert(BT::row_ints() == TypedTables<typename V::Shape, V::THREADS>::row_ints(), "table row layout");
                            btab.ensure((size_t)bgrid.nbricks * BT::row_ints());
                            BrickArgs<real> ta = brick_args();
                            ta.btab = btab.ptr;
                            if (tsub > 1) {                          // the x-quarter boundaries of every brick's (species, tile cell) blocks
                                bsub.ensure((size_t)bgrid.nbricks * BT::NTT + 4);
                                ta.nsub = tsub; ta.bsub = bsub.ptr;
                            }
                            hipLaunchKernelGGL((k_typed_tables<real, typename V::Shape, TT>), dim3(bgrid.per_xcd * NXCD), dim3(TT),
                                               BT::bytes(0), stream(), ta);
                            btab_valid = true;
                            auto kernel = k_typed_build<real, typename V::Shape, V::THREADS, V::GB, V::G>;
                            lds_build_bytes = typed_build_lds_bytes<typename V::Shape, V::THREADS>(tile_cap, own_cap, stride, V::GB, V::G);
                            allow_big_lds(kernel, lds_build_bytes);
                            BrickArgs<real> ba = brick_args();
                            ba.nsub = ta.nsub; ba.bsub = ta.bsub;
                            hipLaunchKernelGGL(kernel, dim3(bgrid.per_xcd * NXCD), dim3(V::THREADS), lds_build_bytes, stream(), ba);
                            return;
                        }
                    }
                    if (!btab_valid && !exp_env("EMDEE_NO_BRICK_TABLES")) {
                        // tables of every brick, once per rebuild; the build and every force launch copy them in
                        // (the image depends on the brick shape only: a small workgroup writes it)
                        constexpr int TT = V::Shape::NTC <= 128 ? 128 : 256;
                        using BT = BrickTables<typename V::Shape, TT>;
                        static_assert(BT::row_ints() == BrickTables<typename V::Shape, V::THREADS>::row_ints(), "table row layout");
                        btab.ensure((size_t)bgrid.nbricks * BT::row_ints());
                        btab_valid = false;
                        BrickArgs<real> ta = brick_args();
                        ta.btab = btab.ptr;
                        ta.stats = (kept && maxima_from_tables) ? reinterpret_cast<unsigned long long *>(flags.ptr + 6) : nullptr;
                        if (nsub > 1) { bsub.ensure((size_t)bgrid.nbricks * V::Shape::NTC + 4); ta.bsub = bsub.ptr; }
                        hipLaunchKernelGGL((k_brick_tables<real, typename V::Shape, TT>), dim3(bgrid.per_xcd * NXCD), dim3(TT),
                                           BT::bytes(0), stream(), ta);
                        btab_valid = true;
                    }
#ifdef EMDEE_EXPERIMENTS
                    if constexpr (std::is_same<V, BrickVariant<0>>::value) {
                        if (tbuild_active<V>()) {
                            auto tk = k_brick_build_t<real, typename V::Shape, V::THREADS, V::G, TB_NPAIR>;
                            lds_build_bytes = brick_tbuild_lds_bytes<typename V::Shape, V::THREADS, TB_NPAIR>(tile_cap, own_cap, stride);
                            allow_big_lds(tk, lds_build_bytes);
                            hipLaunchKernelGGL(tk, dim3(bgrid.per_xcd * NXCD), dim3(V::THREADS), lds_build_bytes, stream(), brick_args());
                            return;
                        }
                    }
#endif
                    auto kernel = k_brick_build<real, typename V::Shape, V::THREADS, V::GB, 1, V::G>;
                    if constexpr (V::GB == 8 || V::GB == 16) {
#ifdef EMDEE_EXPERIMENTS
                        if (build_alg == 2) kernel = k_brick_build<real, typename V::Shape, V::THREADS, V::GB, 2, V::G>;
#endif
                        if (build_alg == 3) kernel = k_brick_build<real, typename V::Shape, V::THREADS, V::GB, 3, V::G>;
                        if (build_alg == 5) kernel = k_brick_build<real, typename V::Shape, V::THREADS, V::GB, 5, V::G>;
                        // round-robin candidates (brick.hpp): when the force kernels read plane values or 16-byte records with 4 lanes per atom
#ifdef EMDEE_EXPERIMENTS
                        constexpr bool STRIDED_G8 = sizeof(real) == 4;   // (8 lanes per atom on 16-byte records: EMDEE_BUILD_STRIDED=1 only)
#else
                        constexpr bool STRIDED_G8 = false;
#endif
                        if constexpr (V::G == 4 || (V::G == 8 && STRIDED_G8)) {
                            const bool strided_ok = (sizeof(real) == 4 || idx_shift != 0) && !exp_env("EMDEE_BUILD_CHUNKED") &&
                                                    (V::G == 4 || exp_env("EMDEE_BUILD_STRIDED") != nullptr);
                            if (build_alg == 3 && strided_ok) kernel = k_brick_build<real, typename V::Shape, V::THREADS, V::GB, 13, V::G>;
#ifdef EMDEE_EXPERIMENTS
                            if constexpr (std::is_same<V, BrickVariant<0>>::value) {
                                if (build_alg == 3 && strided_ok && nsub == 4 && build4_enabled && !build4_blocked && near_far_scale() <= 0.0) {
                                    auto k4 = k_brick_build<real, typename V::Shape, B4_THREADS, B4_G, 13, V::G>;
                                    const size_t lds4 = brick_build_lds_bytes<typename V::Shape, B4_THREADS>(tile_cap, own_cap, stride, B4_G, nsub);
                                    if (lds4 <= LDS_LIMIT) {
                                        lds_build_bytes = lds4;
                                        allow_big_lds(k4, lds4);
                                        launched4 = true;
                                        hipLaunchKernelGGL(k4, dim3(bgrid.per_xcd * NXCD), dim3(B4_THREADS), lds4, stream(), brick_args());
                                        return;
                                    }
                                }
                            }
                            // ... and near entries first (brick.hpp ALG 23), when the skin leaves room for a near radius
                            if (build_alg == 3 && strided_ok && near_far_scale() > 0.0) {
                                kernel = k_brick_build<real, typename V::Shape, V::THREADS, V::GB, 23, V::G>;
                                nearfar_built = true;
                            }
#endif
                            if (build_alg == 5 && strided_ok) kernel = k_brick_build<real, typename V::Shape, V::THREADS, V::GB, 15, V::G>;
                        }
                    }
                    lds_build_bytes = brick_build_lds_bytes<typename V::Shape, V::THREADS>(tile_cap, own_cap, stride, V::GB, nsub);
                    allow_big_lds(kernel, lds_build_bytes);
                    hipLaunchKernelGGL(kernel, dim3(bgrid.per_xcd * NXCD), dim3(V::THREADS), lds_build_bytes, stream(),
                                       brick_args());
                });
            } else {
                nbr.ensure(rows * stride);
                if (n > 0)
                    hipLaunchKernelGGL((k_nbr_build<real>), dim3(blocks_for((size_t)n * WAVE, NBR_BLOCK)), dim3(NBR_BLOCK),
                                       0, stream(), n, n_owned, view(), perm.ptr, cell_sorted.ptr, start(), grid,
                                       (real)(rlist * rlist), nbr.ptr, stride, cnt.ptr, flags.ptr);
            }
            // a build is rare (every ~7 steps): one blocking read-back -- the overflow words and, under a kept plan, the
            // population maxima it has to hold for
            if (in_edit && edit_extra.n > 0) {
                read_back_words(ctx, stream(), flags.ptr, 9, ctx->host_flags, edit_extra.dev, edit_extra.n, edit_extra.host);
                edit_extra_read = true;
            } else {
                read_back_words(ctx, stream(), flags.ptr, 9, ctx->host_flags);
            }
            if (kept && (!plan_holds(ctx->host_flags + 6) || ctx->host_flags[2] != 0)) {
                // the populations outgrew the kept plan (the kernels skipped the bricks concerned): plan afresh and build again
                kept = false;
                plan_valid = false;
                btab_valid = false;
                make_plan();
                continue;
            }
            if (brick_active && ctx->host_flags[4] != 0) {
                // A lane's share of a tile row did not fit its 16-bit hit field.  The 4-lane build (an experiment) sees that
                // when the x sub-bins do not keep a row within 64 slots: this state goes on with 8 lanes per atom.  The default
                // builds cannot (the plan's widest 3-cell run is what picked the field): should they ever, the list is NOT
                // taken -- this state goes on with 32-bit fields.
                if (launched4) {
                    build4_blocked = true;
                    if (std::getenv("EMDEE_DEBUG_PLAN")) std::fprintf(stderr, "emdee plan: a tile row needs %d trips of 4 lanes (> 16): the build goes on with 8 lanes per atom\n", ctx->host_flags[4]);
                    continue;
                }
                EMDEE_REQUIRE(!field16_blocked, EMDEE_ERR_OVERFLOW, "neighbour build: a lane's share of a tile row (%d) overflows its hit field", ctx->host_flags[4]);
                field16_blocked = true;
                kept = false; plan_valid = false; btab_valid = false;
                make_plan();
                continue;
            }
            if (brick_active && ctx->host_flags[3] != 0 && !tbuild_blocked) {
                // a cell with more candidates than the transposed build holds in registers: this state goes on with k_brick_build
                tbuild_blocked = true;
                kept = false;
                plan_valid = false;
                btab_valid = false;
                make_plan();
                continue;
            }
            EMDEE_REQUIRE(ctx->host_flags[2] == 0, EMDEE_ERR_OVERFLOW, "LDS tile overflow (%d records > %d)",
                          ctx->host_flags[2], tile_cap);
            int needed = ctx->host_flags[0];
            if (needed <= stride) {
                builds++;
                has_list = true;
                apply_exclusions();
                return;
            }
            stride = (needed + needed / 8 + 15) / 16 * 16;   // grow and rebuild
            if (brick_active) stride = (stride + row_block - 1) / row_block * row_block;
            if (brick_active && typed_active && !build_fits_lds()) {
                // rows longer than the typed build's LDS row buffers can take: this state goes on with the general-species kernels
                typed_blocked = true;
                plan_valid = false;
                btab_valid = false;
                kept = false;
                make_plan();
                continue;
            }
            if (brick_active && !build_fits_lds()) { brick_active = false; idx_shift = 0; btab_valid = false; plan_valid = false; }
        }
        EMDEE_REQUIRE(false, EMDEE_ERR_OVERFLOW, "neighbour capacity kept overflowing");
    }

    // ---------------------------------------------------------------- exclusions and 1-4 pairs (kernels.hpp)
    // Pairs the caller names (caller ids; bonded neighbours of a molecular model) are struck from the rows right after every
    // build; the 1-4 pairs among them come back scaled (lj14scale of the reference's force-field file, src/modelling.jl:198)
    // through k_pairs14 after every force pass.  Symmetric CSR tables over caller ids, built on the host once per call
    // (topology does not change during a run).  Undivided engines only; two-species boxes with exclusions keep the
    // general-species kernels (a typed row is two block-aligned segments: compacting one would move the other).
    DevBuf<int> ex_start, ex_idx, p14_start, p14_idx;
    bool has_excl = false, has_14 = false;
    double scale14 = 1.0;
    int table_atoms = 0;
    std::vector<int32_t> excl_host, p14_host;          // the caller's pairs as given: {i, j, i, j, ...}
    std::vector<int32_t> fetch_pairs(const int32_t *pairs_dev, int n_pairs) {
        std::vector<int32_t> h((size_t)2 * n_pairs);
        if (n_pairs > 0) {
            EMDEE_HIP_CHECK(hipMemcpyAsync(h.data(), pairs_dev, h.size() * sizeof(int32_t), hipMemcpyDeviceToHost, stream()));
            EMDEE_HIP_CHECK(hipStreamSynchronize(stream()));
        }
        return h;
    }
    // pairs {i, j} of caller ids -> symmetric, sorted, duplicate-free CSR (start[n_atoms + 1], idx) on the device
    void upload_csr(const std::vector<int32_t> &h, int n_atoms, DevBuf<int> &start_out, DevBuf<int> &idx_out) {
        const size_t np = h.size() / 2;
        std::vector<std::pair<int32_t, int32_t>> both;
        both.reserve(2 * np);
        for (size_t k = 0; k < np; k++) {
            const int32_t i = h[2 * k], j = h[2 * k + 1];
            EMDEE_REQUIRE(i >= 0 && j >= 0 && i < n_atoms && j < n_atoms && i != j, EMDEE_ERR_INVALID,
                          "pair table: pair %zu = (%d, %d) is not a pair of two different atoms of %d", k, i, j, n_atoms);
            both.emplace_back(i, j);
            both.emplace_back(j, i);
        }
        std::sort(both.begin(), both.end());
        both.erase(std::unique(both.begin(), both.end()), both.end());
        std::vector<int32_t> st((size_t)n_atoms + 1, 0), ix(both.size());
        for (size_t k = 0; k < both.size(); k++) { st[(size_t)both[k].first + 1]++; ix[k] = both[k].second; }
        for (int a = 0; a < n_atoms; a++) st[(size_t)a + 1] += st[a];
        start_out.ensure(st.size() + 1); idx_out.ensure(ix.size() + 1);
        EMDEE_HIP_CHECK(hipMemcpyAsync(start_out.ptr, st.data(), st.size() * sizeof(int32_t), hipMemcpyHostToDevice, stream()));
        if (!ix.empty()) EMDEE_HIP_CHECK(hipMemcpyAsync(idx_out.ptr, ix.data(), ix.size() * sizeof(int32_t), hipMemcpyHostToDevice, stream()));
        EMDEE_HIP_CHECK(hipStreamSynchronize(stream()));
    }
    // set_excl / set_14: which of the two tables this call replaces (n = 0 clears it); scale: lj14scale
    void set_pair_tables(int n_atoms, const int32_t *excl_dev, int n_excl, bool set_excl, const int32_t *p14_dev, int n_14, bool set_14, double scale) {
        EMDEE_REQUIRE(n_atoms >= 0 && n_excl >= 0 && n_14 >= 0, EMDEE_ERR_INVALID, "pair table: negative count");
        EMDEE_REQUIRE((n_excl == 0 || excl_dev) && (n_14 == 0 || p14_dev), EMDEE_ERR_INVALID, "pair table: NULL array");
        EMDEE_REQUIRE(!set_14 || std::isfinite(scale), EMDEE_ERR_INVALID, "pair table: lj14scale must be finite");
        if (set_excl) excl_host = fetch_pairs(excl_dev, n_excl);
        if (set_14) { p14_host = fetch_pairs(p14_dev, n_14); scale14 = scale; }
        upload_csr(p14_host, n_atoms, p14_start, p14_idx);
        has_14 = !p14_host.empty();
        std::vector<int32_t> all = excl_host;                // what is struck from the rows: the exclusions and the 1-4 pairs together
        all.insert(all.end(), p14_host.begin(), p14_host.end());
        upload_csr(all, n_atoms, ex_start, ex_idx);
        has_excl = !all.empty();
        table_atoms = n_atoms;
        has_list = false; plan_valid = false;                // (the rows in use were filtered with the old tables; typed rows are not filtered)
    }
    // right after a build: the rows without their excluded entries
    void apply_exclusions() {
        if (!has_excl || n_total == 0) return;
        EMDEE_REQUIRE(table_atoms == n_owned && !id_gaps, EMDEE_ERR_STATE, "exclusion tables were set for %d atoms, the state holds %d", table_atoms, n_owned);
        if (brick_active) {
            EMDEE_REQUIRE(!typed_active, EMDEE_ERR_STATE, "exclusions: typed rows are not filtered");
            with_brick_variant(variant, [&](auto v) {
                using V = decltype(v);
                auto kernel = k_brick_filter<real, typename V::Shape, V::THREADS, V::G>;
                using BT = BrickTables<typename V::Shape, V::THREADS>;
                hipLaunchKernelGGL(kernel, dim3(bgrid.per_xcd * NXCD), dim3(V::THREADS), BT::bytes(0), stream(), brick_args(), ex_start.ptr, ex_idx.ptr);
            });
        } else {
            hipLaunchKernelGGL(k_filter_rows, dim3(blocks_for(n_total, 256)), dim3(256), 0, stream(), n_total, n_owned, perm.ptr, nbr.ptr, stride,
                               cnt.ptr, ex_start.ptr, ex_idx.ptr);
        }
    }
    // after a force pass: the scaled 1-4 terms on top
    void add_pairs14(int bitmask) {
        if (!has_14 || n_total == 0) return;
        EMDEE_REQUIRE(table_atoms == n_owned && !id_gaps, EMDEE_ERR_STATE, "1-4 table was set for %d atoms, the state holds %d", table_atoms, n_owned);
        const bool user = brick_active && (out_f || out_e || out_w);
        hipLaunchKernelGGL((k_pairs14<real>), dim3(blocks_for(n_total, 256)), dim3(256), 0, stream(), n_total, n_owned, pitch, view(), perm.ptr,
                           inv_perm.ptr, grid, model, p14_start.ptr, p14_idx.ptr, (real)scale14, bitmask, frc.ptr, en.ptr, vir.ptr,
                           user ? out_f : (real *)nullptr, user ? out_e : (real *)nullptr, user ? out_w : (real *)nullptr);
    }

    // ---------------------------------------------------------------- forces
    const int *direct_guard = nullptr;    // guarded_split_step: the direct kernels of a queued step look at this word first
    template <int BM>
    void launch_direct_force() {
        const int n = n_total;
        int nblocks = (n + FORCE_ATOMS - 1) / FORCE_ATOMS;
        int per_xcd = (nblocks + NXCD - 1) / NXCD;
        hipLaunchKernelGGL((k_lj_force_nbr<real, BM>), dim3(per_xcd * NXCD), dim3(FORCE_BLOCK), 0, stream(), n, n_owned,
                           per_xcd, view(), perm.ptr, nbr.ptr, stride, cnt.ptr, grid, model, pitch, frc.ptr, en.ptr,
                           vir.ptr, direct_guard);
    }

    int force_phase = 0;
    double step_c = 0.0, step_dt = 0.0;
    bool uniform_atoms = false;            // every atom has the same LJAtom (checked when a state is loaded)
    double uni_sigma2 = 0.0, uni_e4 = 0.0, uni_sigma = 1.0;
    // neighbour entries = tile slot << idx_shift; single-species boxes store byte offsets into the coordinate planes
    static constexpr int PLANE_SHIFT = sizeof(real) == 8 ? 3 : 2;
    int idx_shift = 0;

    // Are all LJAtom records identical?  (One small kernel + an 8-byte read-back per load.)
    // uniform_known: -1 = look at the atoms at every load; 0 / 1 = the caller vouches that the species set is mixed /
    // single (emdee_dd_*: agreed once over all domains; atoms only change owner afterwards) and the scan + read-back are skipped
    int uniform_known = -1;
    SpeciesTable species_known{1, {0, 0, 0, 0}};   // with uniform_known == 0: the box's two species, if it has exactly two
    emdee_lj_atom uni_first{0.f, 0.f};
    void detect_uniform_atoms(const emdee_lj_atom *atoms) {
        nt = 1;
        species.n = 1;
        typed_blocked = false;
        tbuild_blocked = false;
        build4_blocked = false;
        field16_blocked = false;
        if (uniform_known >= 0 && n_total > 0) {
            uniform_atoms = uniform_known == 1;
            // (decomposed runs: the two species every domain agreed on at the first load, emdee_dd_load)
            if (!uniform_atoms && species_known.n == 2 && typed_enabled) { species = species_known; nt = 2; }
            return;
        }
        uniform_atoms = false;
        if (n_total == 0 || std::getenv("EMDEE_NO_UNIFORM")) return;
        EMDEE_HIP_CHECK(hipMemsetAsync(flags.ptr + 5, 0, sizeof(int), stream()));
        hipLaunchKernelGGL(k_atoms_differ, dim3(blocks_for(n_total, 256)), dim3(256), 0, stream(), n_total, atoms, flags.ptr + 5);
        // the distinct LJAtom values, if there are few (two: the box is sorted by species and takes the typed kernels)
        unsigned long long tab[MAX_SPECIES + 1];
        // (Float32 operator calls run the reference's arithmetic on the general-species kernels: those read untyped rows)
        const bool want_species = typed_enabled && !(sizeof(real) == 4 && refmath);
        if (want_species) {
            species_tab.ensure(MAX_SPECIES + 1);
            EMDEE_HIP_CHECK(hipMemsetAsync(species_tab.ptr, 0xff, MAX_SPECIES * sizeof(unsigned long long), stream()));
            EMDEE_HIP_CHECK(hipMemsetAsync(species_tab.ptr + MAX_SPECIES, 0, sizeof(unsigned long long), stream()));
            hipLaunchKernelGGL(k_species_collect, dim3(blocks_for(n_total, 256)), dim3(256), 0, stream(), n_total, atoms, species_tab.ptr);
            EMDEE_HIP_CHECK(hipMemcpyAsync(tab, species_tab.ptr, sizeof(tab), hipMemcpyDeviceToHost, stream()));
        }
        emdee_lj_atom first;
        EMDEE_HIP_CHECK(hipMemcpyAsync(&first, atoms, sizeof(first), hipMemcpyDeviceToHost, stream()));
        EMDEE_HIP_CHECK(hipMemcpyAsync(ctx->host_flags + 5, flags.ptr + 5, sizeof(int), hipMemcpyDeviceToHost, stream()));
        EMDEE_HIP_CHECK(hipStreamSynchronize(stream()));
        if (want_species && tab[MAX_SPECIES] == 0) {
            int found = 0;
            for (int q = 0; q < MAX_SPECIES; q++) found += tab[q] != ~0ull ? 1 : 0;
            if (found == 2) {                                    // (the typed kernels are built for two species)
                std::sort(tab, tab + 2);                         // numbering by value, not by who arrived first
                species.n = 2;
                species.key[0] = tab[0]; species.key[1] = tab[1];
                nt = 2;
            }
        }
        uni_first = first;
        if (ctx->host_flags[5] == 0 && first.half_sigma > 0.f && std::isfinite(first.half_sigma)) {
            uniform_atoms = true;
            set_uniform_constants(first);
        }
    }
    // launch constants of the single-species kernels
    void set_uniform_constants(const emdee_lj_atom &first) {
        uni_first = first;
        // the same fp operations as the per-pair path: (hs + hs)^2 and te * te in the kernel's type
        const real sg = (real)first.half_sigma + (real)first.half_sigma;
        uni_sigma = (double)sg;
        uni_sigma2 = (double)(sg * sg);
        uni_e4 = (double)((real)first.twice_sqrt_eps * (real)first.twice_sqrt_eps);
    }

    // ---------------------------------------------------------------- Langevin thermostat (optional)
    // O step between the kick and the drift of every step: v = c1 v + c2 sqrt(T/m) xi(seed, step, id).
    bool lgv_on = false;
    double lgv_gamma = 0.0, lgv_T = 0.0, lgv_c1 = 1.0;
    unsigned long long lgv_seed = 0, lgv_step = 0;
    const long long *lgv_ids = nullptr;    // caller-order ids for the noise counters (NULL: the caller index)
    bool lgv_by_tag = false;               // ... or the tags that travel with the atoms (decomposed domains)
    void set_langevin(double gamma, double temperature, unsigned long long seed, unsigned long long first_step,
                      const long long *ids) {
        EMDEE_REQUIRE(std::isfinite(gamma) && std::isfinite(temperature) && temperature >= 0.0, EMDEE_ERR_INVALID,
                      "langevin: gamma and temperature must be finite, temperature >= 0");
        lgv_on = gamma > 0.0;
        lgv_gamma = gamma; lgv_T = temperature; lgv_seed = seed; lgv_step = first_step; lgv_ids = ids;
    }
    // noise of the step about to be integrated (cell order); the caller advances lgv_step once per step
    void prepare_noise(double dt) {
        if (!lgv_on || n_total == 0) return;
        noise.ensure(3 * pitch);
        lgv_c1 = std::exp(-lgv_gamma * dt);
        const double c2 = std::sqrt(1.0 - lgv_c1 * lgv_c1);
        hipLaunchKernelGGL((k_langevin_noise<real>), dim3(blocks_for(n_total, 256)), dim3(256), 0, stream(), n_total,
                           n_owned, pitch, perm.ptr, lgv_ids, with_mass ? im.ptr : nullptr, lgv_seed, lgv_step, c2, lgv_T,
                           noise.ptr, (use_tags && lgv_by_tag) ? tag.ptr : nullptr);
    }

    // One inner velocity-Verlet step as a single kernel: f(x_k), v += c f/m, x_{k+1} = x_k + dt v written to
    // the other position buffer.  False if the brick kernels are not in use (caller runs the split kernels).
    // guard / trigger (device words, optional): the launch does nothing but raise *trigger when *guard is set, and
    // raises *trigger when an atom it moved is now skin/2 away from its position at the last build (default: flags[1]).
    // carry_ghosts: copy the ghosts' current coordinates into the buffer that becomes current (callers that unpack
    // fresh ghosts before every force evaluation, as emdee_dd_step does, do not need it).
    // noise_ready: the caller has already queued prepare_noise(dt) for this step (it must precede work on another stream).
    bool fused_step(double c, double dt, int phase = 0, const int *guard = nullptr, int *trigger = nullptr,
                    bool carry_ghosts = true, bool noise_ready = false) {
        EMDEE_REQUIRE(has_list && sorted && with_vel, EMDEE_ERR_STATE, "no state loaded");
        if (!brick_active || n_total == 0) return false;
        if (has_14) return false;                            // (the scaled 1-4 terms are added behind a force pass: the split kernels)
        if (phase != 2 && !noise_ready) prepare_noise(dt);   // phases 1 and 2 are the two halves of one step
        {
            Timed t(this, phase == 2 ? T_STEP_BOUNDARY : T_STEP);
            step_c = c; step_dt = dt;
            force_phase = phase;
            step_guard = guard; step_trigger = trigger;
            with_brick_variant(variant, [&](auto v) { launch_brick_kernel<decltype(v), BRICK_STEP, 1>(); });
            step_guard = nullptr; step_trigger = nullptr;
        }
        if (phase != 1 && lgv_on) lgv_step++;
        if (phase != 1) {
            if (carry_ghosts && has_ghosts)
                hipLaunchKernelGGL((k_copy_ghost_records<real>), dim3(blocks_for(n_total, 256)), dim3(256), 0, stream(),
                                   n_total, n_owned, perm.ptr, rec.ptr, rec2.ptr);
            swap_step_buffers();
        }
        return true;
    }
    // positions and velocities ping-pong together
    void swap_step_buffers() { rec.swap(rec2); vel.swap(vel2); }

    // Up to RUN_AHEAD fused steps queued back to back, with ONE read-back for the whole batch instead of a
    // host round trip per step: launch i raises word i when an atom has moved skin/2, launch i + 1 looks at
    // word i first and turns itself into a no-op (passing the word on).  Returns how many steps really ran
    // (>= 1; 0 if the brick kernels are not in use); *stale says whether the last of them asked for a rebuild.
    static constexpr int RUN_AHEAD = 4;
    int run_ahead = RUN_AHEAD;            // EMDEE_RUN_AHEAD=1: one step per round trip (profiling: no no-op launches)
    int *step_trigger = nullptr;
    const int *step_guard = nullptr;
    int fused_steps_run_ahead(double c, double dt, int want, bool *stale) {
        EMDEE_REQUIRE(has_list && sorted && with_vel, EMDEE_ERR_STATE, "no state loaded");
        *stale = false;
        if (!brick_active || n_total == 0 || has_ghosts || has_14) return 0;
        const int B = std::max(1, std::min(want, run_ahead));
        int *words = flags.ptr + 9;                          // flags[9 .. 9 + RUN_AHEAD)
        EMDEE_HIP_CHECK(hipMemsetAsync(words, 0, B * sizeof(int), stream()));
        step_c = c; step_dt = dt; force_phase = 0;
        for (int i = 0; i < B; i++) {
            prepare_noise(dt);
            if (lgv_on) lgv_step++;
            Timed t(this, T_STEP);
            step_trigger = words + i;
            step_guard = i ? words + i - 1 : nullptr;
            with_brick_variant(variant, [&](auto v) { launch_brick_kernel<decltype(v), BRICK_STEP, 1>(); });
            swap_step_buffers();
        }
        step_trigger = nullptr; step_guard = nullptr;
        read_back_words(ctx, stream(), words, B, ctx->host_flags + 9);
        int ran = B;
        for (int i = 0; i < B; i++)
            if (ctx->host_flags[9 + i]) { ran = i + 1; *stale = true; break; }
        if ((B - ran) & 1) swap_step_buffers();              // the skipped launches did not advance the ping-pong
        if (lgv_on) lgv_step -= (unsigned long long)(B - ran);
        if (profiling) timers[T_STEP].dropped += B - ran;
        return ran;
    }

    // operator path: outputs of the next compute_forces go straight to these caller-order arrays (tiled kernels only)
    real *out_f = nullptr, *out_e = nullptr, *out_w = nullptr;
    // fp32 operator path: pair geometry in the reference's own Float32 arithmetic (scaled positions, minimum image per
    // pair; brick.hpp BrickArgs::refmath) -- what keeps compute_nonbonded! within the reference's 1e-4 of its CPU loop
    bool refmath = false;
    const real *ref_pos = nullptr;        // the caller's positions of the current operator call (refmath tiles are staged from them)

    void compute_forces(int bitmask, int phase = 0) {
        EMDEE_REQUIRE(has_list, EMDEE_ERR_STATE, "no neighbour list");
        EMDEE_REQUIRE(bitmask >= 0 && bitmask <= 7, EMDEE_ERR_INVALID, "bitmask must be a combination of 1|2|4");
        if (n_total == 0 || bitmask == 0) return;
        if (!brick_active && phase == 1) return;   // the direct kernels have no brick phases: all work in phase 2
        Timed t(this, T_FORCE);
        force_phase = brick_active ? phase : 0;
        if (brick_active) {
            with_brick_variant(variant, [&](auto v) { launch_brick_force<decltype(v)>(bitmask); });
            add_pairs14(bitmask);
            return;
        }
        switch (bitmask) {
            case 1: launch_direct_force<1>(); break;
            case 2: launch_direct_force<2>(); break;
            case 3: launch_direct_force<3>(); break;
            case 4: launch_direct_force<4>(); break;
            case 5: launch_direct_force<5>(); break;
            case 6: launch_direct_force<6>(); break;
            default: launch_direct_force<7>(); break;
        }
        add_pairs14(bitmask);
    }

    // ---------------------------------------------------------------- integrator
    void kick_drift(double c, double dt, int *trigger = nullptr, const int *guard = nullptr) {
        EMDEE_REQUIRE(sorted && with_vel, EMDEE_ERR_STATE, "no velocities loaded");
        if (n_total == 0) return;
        prepare_noise(dt);
        Timed t(this, T_KICK_DRIFT);
        real thr = (real)(0.5 * skin);
        hipLaunchKernelGGL((k_kick_drift<real>), dim3(blocks_for(n_total, 256)), dim3(256), 0, stream(), n_total, n_owned,
                           pitch, perm.ptr, rec.ptr, vel.ptr, frc.ptr, with_mass ? im.ptr : nullptr, (real)c, (real)dt,
                           xb.ptr, thr * thr, trigger ? trigger : flags.ptr + 1, lgv_on ? noise.ptr : nullptr, (real)lgv_c1, guard,
                           far_skip_active() ? flags.ptr + 16 : (int *)nullptr, (real)(0.25 * near_delta() * near_delta()));
        if (lgv_on) lgv_step++;
    }

    // One inner step of a domain whose tiles do not fit LDS (the direct kernels), obeying the same device words as
    // fused_step: nothing happens if *guard is set (and *trigger is raised, passing the request on), otherwise force pass,
    // full kick and drift in place, *trigger raised if an atom is now skin/2 away from its position at the last build.
    // Keeps a decomposed run's message sequence independent of which kernels a domain uses (emdee_dd_step).
    void guarded_split_step(double c, double dt, const int *guard, int *trigger) {
        EMDEE_REQUIRE(has_list && sorted && with_vel, EMDEE_ERR_STATE, "no state loaded");
        EMDEE_REQUIRE(!brick_active, EMDEE_ERR_STATE, "guarded_split_step is the direct kernels' form of fused_step");
        if (n_total == 0) return;
        direct_guard = guard;
        compute_forces(EMDEE_FORCES, 0);
        direct_guard = nullptr;
        kick_drift(c, dt, trigger, guard);
    }

    void kick(double c) {
        EMDEE_REQUIRE(sorted && with_vel, EMDEE_ERR_STATE, "no velocities loaded");
        if (n_total == 0) return;
        Timed t(this, T_KICK);
        hipLaunchKernelGGL((k_kick<real>), dim3(blocks_for(n_total, 256)), dim3(256), 0, stream(), n_total, n_owned, pitch,
                           perm.ptr, vel.ptr, frc.ptr, with_mass ? im.ptr : nullptr, (real)c);
    }

    // blocking read of the rebuild trigger raised by kick_drift / check_user_displacement
    bool read_rebuild_flag() {
        read_back_words(ctx, stream(), flags.ptr + 1, 1, ctx->host_flags + 1);
        return ctx->host_flags[1] != 0;
    }

    // ---------------------------------------------------------------- observables
    // out[0] = sum e, out[1] = kinetic energy (velocities advanced by a pending c f/m), out[2] = sum w
    void energy_sums(double pending_c, double out[3]) {
        out[0] = out[1] = out[2] = 0.0;
        if (n_total == 0) return;
        int nb = std::min((int)blocks_for(n_total, RED_BLOCK), RED_MAX_BLOCKS);
        hipLaunchKernelGGL((k_energy_partials<real>), dim3(nb), dim3(RED_BLOCK), 0, stream(), n_total, n_owned, pitch,
                           perm.ptr, en.ptr, vir.ptr, with_vel ? vel.ptr : nullptr, frc.ptr,
                           with_mass ? im.ptr : nullptr, (real)pending_c, partial.ptr);
        hipLaunchKernelGGL(k_final_sum3, dim3(1), dim3(RED_BLOCK), 0, stream(), nb, partial.ptr, sums.ptr);
        EMDEE_HIP_CHECK(hipMemcpyAsync(out, sums.ptr, 3 * sizeof(double), hipMemcpyDeviceToHost, stream()));
        EMDEE_HIP_CHECK(hipStreamSynchronize(stream()));
    }

    void list_stats(bool count_pairs, int64_t *listed, int32_t *max_count, int64_t *inside) {
        unsigned long long h[3] = {0, 0, 0};
        if (has_list && n_total > 0) {
            EMDEE_HIP_CHECK(hipMemsetAsync(stats.ptr, 0, 3 * sizeof(unsigned long long), stream()));
            if (brick_active) {
                with_brick_variant(variant, [&](auto v) { launch_brick_kernel<decltype(v), BRICK_STATS, 0>(); });
            } else {
                int nb = std::min((int)blocks_for(n_total, RED_BLOCK), RED_MAX_BLOCKS);
                hipLaunchKernelGGL((k_list_stats<real>), dim3(nb), dim3(RED_BLOCK), 0, stream(), n_total, view(), nbr.ptr,
                                   stride, cnt.ptr, grid, model.rc2, count_pairs ? 1 : 0, stats.ptr);
            }
            EMDEE_HIP_CHECK(hipMemcpyAsync(h, stats.ptr, sizeof(h), hipMemcpyDeviceToHost, stream()));
            EMDEE_HIP_CHECK(hipStreamSynchronize(stream()));
        }
        if (listed) *listed = (int64_t)h[0];
        if (max_count) *max_count = (int32_t)h[1];
        if (inside) *inside = (int64_t)(h[2] / 2);   // full list: every pair appears twice
    }

    // neighbour rows as caller ids (verification accessor): counts[n_owned], out[n_owned x capacity]
    void export_list(int *counts, int *out, int capacity) {
        EMDEE_REQUIRE(has_list, EMDEE_ERR_STATE, "no neighbour list");
        EMDEE_REQUIRE(counts && out && capacity > 0, EMDEE_ERR_INVALID, "export_list: bad arguments");
        if (n_total == 0) return;
        const int *map = ids_map();
        if (brick_active) {
            with_brick_variant(variant, [&](auto v) {
                using V = decltype(v);
                if constexpr (typed_variant<V>()) {
                    if (typed_active) {
                        auto tk = k_typed_export<real, typename V::Shape, V::THREADS, V::G>;
                        using TTab = TypedTables<typename V::Shape, V::THREADS>;
                        const size_t tlds = TTab::bytes(0);
                        hipLaunchKernelGGL(tk, dim3(bgrid.per_xcd * NXCD), dim3(V::THREADS), tlds, stream(), brick_args(), counts, out, capacity, map);
                        return;
                    }
                }
                auto kernel = k_brick_export<real, typename V::Shape, V::THREADS, V::G>;
                using BT = BrickTables<typename V::Shape, V::THREADS>;
                const size_t lds = BT::bytes(0);
                hipLaunchKernelGGL(kernel, dim3(bgrid.per_xcd * NXCD), dim3(V::THREADS), lds, stream(), brick_args(), counts, out,
                                   capacity, map);
            });
        } else {
            hipLaunchKernelGGL(k_export_rows, dim3(blocks_for(n_total, 256)), dim3(256), 0, stream(), n_total, n_owned, perm.ptr, nbr.ptr,
                               stride, cnt.ptr, counts, out, capacity, map);
        }
        EMDEE_HIP_CHECK(hipGetLastError());
    }

    // ---------------------------------------------------------------- caller-order copies
    // (dense caller order: owned atoms first, then the ghosts -- through ids_map() when the ids have gaps; atoms_out / tags_out
    // / raw: what a decomposition needs to go back to caller-order arrays, dd.hpp)
    void unsort(real *pos, real *velocities, real *forces, real *energies, real *virials, emdee_lj_atom *atoms_out = nullptr,
                long long *tags_out = nullptr, bool raw = false) {
        if (n_total == 0) return;
        EMDEE_REQUIRE(!velocities || with_vel, EMDEE_ERR_STATE, "no velocities loaded");
        EMDEE_REQUIRE(!tags_out || use_tags, EMDEE_ERR_STATE, "no tags loaded");
        const int *map = ids_map();
        hipLaunchKernelGGL((k_unsort<real>), dim3(blocks_for(n_total, 256)), dim3(256), 0, stream(), n_owned, n_total, pitch,
                           grid, img.ptr, perm.ptr, map, rec.ptr, te.ptr, with_vel ? vel.ptr : nullptr, frc.ptr, en.ptr, vir.ptr,
                           use_tags ? tag.ptr : nullptr, pos, velocities, forces, energies, virials, atoms_out, tags_out, raw ? 1 : 0,
                           rel_grid(rel_now, cell_sorted.ptr));
    }

    // operator path: does the cached list still cover these caller positions?
    bool user_positions_moved(const real *pos) {
        if (n_total == 0) return false;
        real thr = (real)(0.5 * skin);
        EMDEE_HIP_CHECK(hipMemsetAsync(flags.ptr + 1, 0, sizeof(int), stream()));
        hipLaunchKernelGGL((k_check_displacement<real>), dim3(blocks_for(n_total, 256)), dim3(256), 0, stream(), n_total,
                           inv_perm.ptr, pos, xb.ptr, pitch, grid, thr * thr, flags.ptr + 1);
        return read_rebuild_flag();
    }

    // operator path, list kept: refresh + displacement test + species test in one pass and one read-back.
    // Returns true if the list no longer covers the positions (the caller reloads).
    bool refresh_and_check(const real *pos, const emdee_lj_atom *atoms) {
        if (n_total == 0) return false;
        const real thr = (real)(0.5 * skin);
        EMDEE_HIP_CHECK(hipMemsetAsync(flags.ptr + 1, 0, sizeof(int), stream()));
        EMDEE_HIP_CHECK(hipMemsetAsync(flags.ptr + 5, 0, sizeof(int), stream()));
        hipLaunchKernelGGL((k_refresh_check<real>), dim3(blocks_for(n_total, 256)), dim3(256), 0, stream(), n_total, pitch,
                           grid, perm.ptr, pos, atoms, xb.ptr, rec.ptr, te.ptr, thr * thr, flags.ptr, nt > 1 ? 1 : 0);
        emdee_lj_atom first;
        read_back_words(ctx, stream(), flags.ptr, 16, ctx->host_flags);        // posted: 6 us of empty queue instead of two copies + a sync
        memcpy(&first.half_sigma, &ctx->host_flags[14], 4);
        memcpy(&first.twice_sqrt_eps, &ctx->host_flags[15], 4);
        uniform_atoms = false;
        uni_first = first;                                   // (what a decomposition votes on, and what the next load starts from)
        if (ctx->host_flags[5] == 0 && !std::getenv("EMDEE_NO_UNIFORM") && first.half_sigma > 0.f && std::isfinite(first.half_sigma)) {
            uniform_atoms = true;
            set_uniform_constants(first);
        }
        return ctx->host_flags[1] != 0;
    }

    void refresh_user(const real *pos, const emdee_lj_atom *atoms) {
        if (n_total == 0) return;
        hipLaunchKernelGGL((k_refresh_positions<real>), dim3(blocks_for(n_total, 256)), dim3(256), 0, stream(), n_total,
                           pitch, grid, perm.ptr, pos, atoms, xb.ptr, rec.ptr, te.ptr);
    }
};

}  // namespace emdee
