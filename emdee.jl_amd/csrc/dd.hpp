// dd.hpp -- spatial domain decomposition behind the C ABI (emdee_dd_*).
//
// One emdee_dd object drives the domains that live in THIS process:
//   * production: one process per GPU, one domain per process, messages over RCCL (xGMI) -- ncclSend/ncclRecv
//     grouped per step on a communication stream of its own, resolved from librccl at run time;
//   * validation on a one-GPU box: all domains of the decomposition in one process on one device, messages as
//     device-to-device copies between the domains' streams (same code path up to the transport call).
//
// A velocity-Verlet step of a decomposed box is, per domain and with no host round trip:
//   pack ghost-source positions (+ this domain's rebuild request) -> exchange, in flight on the communication
//   stream || fused force + kick + drift over the INTERIOR bricks (their LDS tile holds no ghost cell)
//   -> unpack ghosts, OR of all requests -> the same kernel over the BOUNDARY bricks.
// The rebuild decision rides on the halo messages: with at most three bricks per dimension every domain is a
// neighbour of every other one, so the requests of a step reach everybody with the positions they refer to.
// A step whose positions were flagged does nothing (device-side guard words), as do all steps queued behind
// it; the host reads the words back once per batch of queued steps, rewinds to the flagged step and rebuilds:
// migration (only atoms that left their brick travel), new ghost lists, re-sort, neighbour list.
#pragma once

#include <dlfcn.h>

#include <memory>
#include <vector>

#include "dd_kernels.hpp"
#include "impl.hpp"

namespace emdee {

// ------------------------------------------------------------------------------------ RCCL, resolved at run time
// (libemdee_hip.so carries no link-time dependency on librccl: single-GPU users never load it, and inside a
// PyTorch process the already-loaded librccl.so.1 is the one that is found)
struct RcclApi {
    typedef struct { char internal[128]; } UniqueId;
    typedef void *Comm;
    void *handle = nullptr;
    int (*GetUniqueId)(UniqueId *) = nullptr;
    int (*CommInitRank)(Comm *, int, UniqueId, int) = nullptr;
    int (*CommDestroy)(Comm) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    int (*Send)(const void *, size_t, int, int, Comm, hipStream_t) = nullptr;
    int (*Recv)(void *, size_t, int, int, Comm, hipStream_t) = nullptr;
    int (*AllReduce)(const void *, void *, size_t, int, int, Comm, hipStream_t) = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
    static constexpr int kChar = 0, kFloat64 = 8, kSum = 0;   // ncclInt8 / ncclFloat64 / ncclSum (rccl.h)

    static RcclApi &get() {
        static RcclApi api;
        return api;
    }
    void load() {
        if (handle) return;
        const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char *n : names) {
            handle = dlopen(n, RTLD_NOW | RTLD_LOCAL);
            if (handle) break;
        }
        EMDEE_REQUIRE(handle != nullptr, EMDEE_ERR_INVALID, "librccl.so.1 not found (%s)", dlerror());
        auto sym = [&](const char *name) {
            void *p = dlsym(handle, name);
            EMDEE_REQUIRE(p != nullptr, EMDEE_ERR_INVALID, "librccl: missing symbol %s", name);
            return p;
        };
        GetUniqueId = reinterpret_cast<decltype(GetUniqueId)>(sym("ncclGetUniqueId"));
        CommInitRank = reinterpret_cast<decltype(CommInitRank)>(sym("ncclCommInitRank"));
        CommDestroy = reinterpret_cast<decltype(CommDestroy)>(sym("ncclCommDestroy"));
        GroupStart = reinterpret_cast<decltype(GroupStart)>(sym("ncclGroupStart"));
        GroupEnd = reinterpret_cast<decltype(GroupEnd)>(sym("ncclGroupEnd"));
        Send = reinterpret_cast<decltype(Send)>(sym("ncclSend"));
        Recv = reinterpret_cast<decltype(Recv)>(sym("ncclRecv"));
        AllReduce = reinterpret_cast<decltype(AllReduce)>(sym("ncclAllReduce"));
        GetErrorString = reinterpret_cast<decltype(GetErrorString)>(sym("ncclGetErrorString"));
    }
};
#define EMDEE_RCCL_CHECK(expr)                                                                       \
    do {                                                                                             \
        int r_ = (expr);                                                                             \
        if (r_ != 0) {                                                                               \
            ::emdee::set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr,                         \
                               ::emdee::RcclApi::get().GetErrorString(r_));                          \
            throw ::emdee::Failure{EMDEE_ERR_HIP};                                                   \
        }                                                                                            \
    } while (0)

// ------------------------------------------------------------------------------------ geometry (host)
struct DdGeom {
    double L[3], halo;
    int grid[3], world;
    int rank, coords[3];
    double width[3], lo[3], hi[3], local_lo[3], local_len[3];
    int cut[3], periodic[3];
    int ndirs = 0;
    int dir[DD_MAX_DIRS][3], dir_rank[DD_MAX_DIRS];
    double dir_shift[DD_MAX_DIRS][3];
    int npeers = 0;
    int peers[DD_MAX_PEERS];             // distinct neighbour ranks, ascending
    int ghost_nbins = 0;
    int dir_of_bin[DD_MAX_DIRS];         // send-list bins: directions sorted by (destination rank, direction)
    int bin_of_dir[DD_MAX_DIRS];
    int peer_bin_lo[DD_MAX_PEERS + 2];   // bins of peer p: [peer_bin_lo[p], peer_bin_lo[p+1])
    bool mirror = false;                 // DdImpl::mirror: every peer is this domain's own periodic image

    int peer_index(int r) const {
        for (int p = 0; p < npeers; p++)
            if (peers[p] == r) return p;
        return -1;
    }

    void init(const double L_[3], const int grid_[3], double halo_, int rank_) {
        halo = halo_; rank = rank_;
        world = grid_[0] * grid_[1] * grid_[2];
        for (int d = 0; d < 3; d++) { L[d] = L_[d]; grid[d] = grid_[d]; }
        coords[0] = rank % grid[0]; coords[1] = (rank / grid[0]) % grid[1]; coords[2] = rank / (grid[0] * grid[1]);
        for (int d = 0; d < 3; d++) {
            width[d] = L[d] / grid[d];
            lo[d] = coords[d] * width[d];
            hi[d] = lo[d] + width[d];
            cut[d] = grid[d] > 1;
            EMDEE_REQUIRE(grid[d] <= 3, EMDEE_ERR_INVALID,
                          "emdee_dd: at most 3 bricks per dimension (every domain must neighbour every other one)");
            if (cut[d]) EMDEE_REQUIRE(halo <= width[d], EMDEE_ERR_INVALID, "halo %g exceeds the brick width %g along dimension %d", halo, width[d], d);
            else EMDEE_REQUIRE(2.0 * halo <= L[d], EMDEE_ERR_INVALID, "cutoff + skin exceeds half the periodic length along dimension %d", d);
            local_lo[d] = cut[d] ? lo[d] - halo : 0.0;
            local_len[d] = cut[d] ? width[d] + 2 * halo : L[d];
            periodic[d] = cut[d] ? 0 : 1;
        }
        // 26 directions in the fixed order of domain.py (x fastest), only cut dimensions may be non-zero
        ndirs = 0;
        for (int sz = -1; sz <= 1; sz++)
            for (int sy = -1; sy <= 1; sy++)
                for (int sx = -1; sx <= 1; sx++) {
                    const int s[3] = {sx, sy, sz};
                    if (sx == 0 && sy == 0 && sz == 0) continue;
                    bool ok = true;
                    for (int d = 0; d < 3; d++) ok = ok && (s[d] == 0 || cut[d]);
                    if (!ok) continue;
                    int n[3];
                    for (int d = 0; d < 3; d++) {
                        const int c = coords[d] + s[d];
                        dir[ndirs][d] = s[d];
                        dir_shift[ndirs][d] = c >= grid[d] ? -L[d] : (c < 0 ? L[d] : 0.0);
                        n[d] = ((c % grid[d]) + grid[d]) % grid[d];
                    }
                    dir_rank[ndirs] = n[0] + grid[0] * (n[1] + grid[1] * n[2]);
                    ndirs++;
                }
        npeers = 0;
        for (int r = 0; r < world; r++) {
            if (r == rank) continue;
            for (int k = 0; k < ndirs; k++)
                if (dir_rank[k] == r) { peers[npeers++] = r; break; }
        }
        // a rank may be its own neighbour only along a dimension with one brick, and those are not cut
        for (int k = 0; k < ndirs; k++) EMDEE_REQUIRE(dir_rank[k] != rank, EMDEE_ERR_INVALID, "emdee_dd: a brick neighbours itself");
        ghost_nbins = 0;
        for (int p = 0; p < npeers; p++) {
            peer_bin_lo[p] = ghost_nbins;
            for (int k = 0; k < ndirs; k++)
                if (dir_rank[k] == peers[p]) { dir_of_bin[ghost_nbins] = k; bin_of_dir[k] = ghost_nbins; ghost_nbins++; }
        }
        peer_bin_lo[npeers] = ghost_nbins;
        peer_bin_lo[npeers + 1] = ghost_nbins;
    }

    template <typename real>
    DdDev<real> device() const {
        DdDev<real> g{};
        for (int d = 0; d < 3; d++) {
            g.L[d] = (real)L[d]; g.width[d] = (real)width[d]; g.lo[d] = (real)lo[d]; g.hi[d] = (real)hi[d];
            g.grid[d] = grid[d]; g.cut[d] = cut[d];
        }
        g.halo = (real)halo; g.rank = rank; g.world = world; g.ndirs = ndirs;
        for (int k = 0; k < ndirs; k++) {
            for (int d = 0; d < 3; d++) { g.dir[k][d] = dir[k][d]; g.shift[k][d] = (real)dir_shift[k][d]; }
            g.dir_bin[k] = bin_of_dir[k];
            g.bin_dir[bin_of_dir[k]] = k;
        }
        for (int r = 0; r < DD_MAX_WORLD; r++) g.rank_bin[r] = -1;
        g.rank_bin[rank] = 0;
        for (int p = 0; p < npeers; p++) g.rank_bin[peers[p]] = 1 + p;
        for (int b = 0; b <= DD_MAX_PEERS; b++)
            for (int d = 0; d < 3; d++) g.mig_off[b][d] = (real)0;
        if (mirror) {
            // The replica rehearsal: what a peer sends me is what I send it, seen from my side.  An atom of mine in the halo of
            // my face in direction s is the peer's atom in the halo of ITS face in direction s (grids of <= 2 bricks per
            // dimension: s reaches the same peer both ways), which the peer sends me as a ghost on my OTHER side -- my atom
            // moved by -s times the brick width; an atom that leaves me for the brick at coordinates c arrives from there in my
            // own brick, moved by -(c - mine) widths.
            for (int k = 0; k < ndirs; k++)
                for (int d = 0; d < 3; d++) g.shift[k][d] = (real)(-dir[k][d] * width[d]);
            for (int p = 0; p < npeers; p++) {
                const int r = peers[p];
                const int c[3] = {r % grid[0], (r / grid[0]) % grid[1], r / (grid[0] * grid[1])};
                for (int d = 0; d < 3; d++) g.mig_off[1 + p][d] = (real)((c[d] - coords[d]) * width[d]);
            }
        }
        return g;
    }
};

// what one domain hands to a transport call: one message per peer, both ways
struct Xfer {
    const unsigned char *send = nullptr;
    unsigned char *recv = nullptr;
    size_t soff[DD_MAX_PEERS], sbytes[DD_MAX_PEERS], roff[DD_MAX_PEERS], rbytes[DD_MAX_PEERS];
};

constexpr int DD_MAX_BATCH = 8;
constexpr int DD_WORDS = 2 * (DD_MAX_BATCH + 2);   // V[0..MAXB+1] | G[0..MAXB+1]

// ------------------------------------------------------------------------------------ one domain
template <typename real>
struct Domain {
    DdGeom geo;
    emdee_ctx *ctx = nullptr;            // compute stream (+ pinned words)
    bool owns_ctx = false;
    hipStream_t comm = nullptr;          // communication stream
    hipStream_t side = nullptr;          // boundary bricks: unpack + second launch of a step, concurrent with the interior launch's tail
    hipEvent_t ev_packed = nullptr, ev_done = nullptr, ev_bnd = nullptr;
    bool halo_timed = false;             // T_HALO pair of the step in flight (in-order / in-process forms)
    bool halo_batch_timed = false;       // ... of any step of the batch in flight
    size_t halo_tk = 0;
    std::unique_ptr<MdImpl<real>> md;
    // caller-order working arrays (owned first, then ghosts for x and atoms), double-buffered across a migration
    DevBuf<real> x, x2, v, v2, f;
    DevBuf<emdee_lj_atom> at, at2;
    DevBuf<long long> gid, gid2;
    DevBuf<unsigned> mask, gmask;
    DevBuf<unsigned char> keep;                    // rebuild in the engine's order: slots that make it into the new state
    DevBuf<int> counts, counts2, ids, bins, codes, small;   // small: bin_start[33] | cnt_send[27] | cnt_recv[27] | err[1]
    DevBuf<unsigned char> sendbuf, recvbuf;
    DevBuf<int> words;
    DevBuf<double> red;
    Scanner scanner;
    int n_owned = 0, n_ghost = 0, n_send = 0;
    int host_small[96];
    // count-free rebuilds (dd_kernels.hpp): capacities both ends of a message know, the words of the one read-back
    DdCaps mig_caps{}, gs_caps{}, gr_caps{};
    bool have_caps = false;
    DevBuf<int> w;
    int host_w[DDW_COUNT];
    DdPlan plan{};
    Xfer xf{};
    int since_build = 0;
    bool words_clear = false;            // the guard words were cleared by the rebuild that has just run

    int *V(int j) { return words.ptr + j; }
    int *G(int j) { return words.ptr + (DD_MAX_BATCH + 2) + j; }
    hipStream_t stream() const { return ctx->stream; }
    NbSystem<real> &sys() { return md->sys; }
};

// ------------------------------------------------------------------------------------ the decomposition
template <typename real>
struct DdImpl : IDd {
    emdee_ctx *user_ctx;
    double L[3], skin, halo;
    int grid[3], world;
    emdee_lj_model model;
    std::vector<std::unique_ptr<Domain<real>>> dom;   // local domains, ranks rank_first .. rank_first + n_local - 1
    int rank_first, n_local;
    bool use_rccl = false;
    // The replica rehearsal of ONE rank on one device (n_local = 1 of a larger grid, no communicator id): every peer is taken
    // to hold this domain's own atoms, moved by whole brick widths -- the box is then periodic with the BRICK's period, and
    // what a peer would send is what this domain sends it, seen from this side (DdGeom::device).  Transport: a copy of the
    // send buffer's messages into the receive buffer.  Everything else -- seven peers, 26 directions, migration, ghost
    // selection, padded messages, the per-step halo on its own stream, batches and guard words -- is the production path
    // with the production message sizes: the cost of one rank of an N-rank run, minus the links, measured on one GPU; and
    // the run must reproduce the undivided periodic box of one brick (tests/test_gpu_dd.py).
    bool mirror = false;
    RcclApi::Comm comm = nullptr;
    bool loaded = false;
    int64_t n_global = 0;
    int max_batch = DD_MAX_BATCH;
    bool overlap = true;
    bool hold_interior = std::getenv("EMDEE_DD_HOLD_INTERIOR") == nullptr || std::atoi(std::getenv("EMDEE_DD_HOLD_INTERIOR")) != 0;
    bool two_streams = true;              // EMDEE_DD_STREAMS=1: interior and boundary launches on one stream
    bool no_shortcut = false;             // EMDEE_DD_NO_SHORTCUT=1: a one-domain grid goes through the whole ownership path (profiling)
    bool count_free = true;               // EMDEE_DD_COUNT_FREE=0: every rebuild exchanges its row counts first (round 2)
    bool lockstep = true;                 // EMDEE_DD_LOCKSTEP=0: the overlapped form on three streams (round 2)
    bool halo_live = false;               // the halo stream holds work the compute stream has not waited for yet
    int mig_cap_forced = 0;               // EMDEE_DD_MIG_CAP=n: migrant rows per message (tests: 1 forces the redo with counts)
    bool ghost_cap_exact = false;         // EMDEE_DD_GHOST_CAP=exact: ghost messages hold the rows of the previous rebuild and no more (tests)
    int64_t stat_fast = 0, stat_fallback = 0;
    int last_interval = 0;
    // Langevin
    bool lgv_on = false;
    double lgv_gamma = 0, lgv_T = 0;
    uint64_t lgv_seed = 0, lgv_first = 0;
    int64_t stat_batches = 0, stat_cancelled = 0, stat_rebuilds = 0, stat_migrated = 0;

    DdImpl(emdee_ctx *c, const double len[3], const int g[3], int rank_first_, int n_local_, const void *unique_id,
           const emdee_lj_model &m, double skin_)
        : user_ctx(c), skin(skin_), model(m), rank_first(rank_first_), n_local(n_local_) {
        // (a constructor that throws runs no destructor: streams, events, pinned words and the communicator created so far
        // are released here, e.g. when ncclCommInitRank fails because two ranks share a device)
        try {
            construct(len, g, unique_id);
        } catch (...) {
            release();
            throw;
        }
    }

    void construct(const double len[3], const int g[3], const void *unique_id) {
        emdee_ctx *c = user_ctx;
        const emdee_lj_model &m = model;
        use_device(c);
        for (int d = 0; d < 3; d++) {
            L[d] = len[d]; grid[d] = g[d];
            EMDEE_REQUIRE(g[d] >= 1 && len[d] > 0, EMDEE_ERR_INVALID, "emdee_dd: grid and box lengths must be positive");
        }
        world = g[0] * g[1] * g[2];
        EMDEE_REQUIRE(world <= DD_MAX_WORLD, EMDEE_ERR_INVALID, "emdee_dd: at most %d domains", DD_MAX_WORLD);
        EMDEE_REQUIRE(n_local >= 1 && rank_first >= 0 && rank_first + n_local <= world, EMDEE_ERR_INVALID,
                      "emdee_dd: local ranks [%d, %d) outside the %d-domain grid", rank_first, rank_first + n_local, world);
        EMDEE_REQUIRE(m.rc2 > 0 && skin >= 0, EMDEE_ERR_INVALID, "emdee_dd: bad model or skin");
        halo = std::sqrt(m.rc2) + skin;
        use_rccl = n_local < world && unique_id != nullptr;
        mirror = n_local < world && unique_id == nullptr;
        if (n_local < world) EMDEE_REQUIRE(n_local == 1, EMDEE_ERR_INVALID, "emdee_dd: one domain per process when the domains span processes");
        if (mirror)
            for (int d = 0; d < 3; d++)
                EMDEE_REQUIRE(g[d] <= 2, EMDEE_ERR_INVALID, "emdee_dd: the replica rehearsal (one local domain, no communicator id) needs at most 2 bricks per dimension");
        if (const char *e = std::getenv("EMDEE_DD_BATCH")) max_batch = std::max(1, std::min(DD_MAX_BATCH, std::atoi(e)));
        if (const char *e = std::getenv("EMDEE_DD_OVERLAP")) overlap = std::atoi(e) != 0;
        if (const char *e = std::getenv("EMDEE_DD_STREAMS")) two_streams = std::atoi(e) != 1;
        if (const char *e = std::getenv("EMDEE_DD_NO_SHORTCUT")) no_shortcut = std::atoi(e) != 0;
        if (const char *e = std::getenv("EMDEE_DD_COUNT_FREE")) count_free = std::atoi(e) != 0;
        if (const char *e = std::getenv("EMDEE_DD_LOCKSTEP")) lockstep = std::atoi(e) != 0;
        if (const char *e = std::getenv("EMDEE_DD_MIG_CAP")) mig_cap_forced = std::max(1, std::atoi(e));
        if (const char *e = std::getenv("EMDEE_DD_GHOST_CAP")) ghost_cap_exact = std::string(e) == "exact";
        for (int l = 0; l < n_local; l++) {
            dom.push_back(std::make_unique<Domain<real>>());   // registered first: release() sees whatever it gets below
            Domain<real> *d = dom.back().get();
            d->geo.init(L, grid, halo, rank_first + l);
            d->geo.mirror = mirror;
            if (use_rccl || mirror) {
                d->ctx = c;
            } else {
                d->ctx = new emdee_ctx(*c);
                d->owns_ctx = true;
                d->ctx->stream = nullptr;
                d->ctx->host_flags = nullptr;
                EMDEE_HIP_CHECK(hipStreamCreateWithFlags(&d->ctx->stream, hipStreamNonBlocking));
                d->ctx->post_dev = nullptr;
                host_words_alloc(d->ctx);
            }
            EMDEE_HIP_CHECK(hipStreamCreateWithFlags(&d->comm, hipStreamNonBlocking));
            EMDEE_HIP_CHECK(hipStreamCreateWithFlags(&d->side, hipStreamNonBlocking));
            EMDEE_HIP_CHECK(hipEventCreateWithFlags(&d->ev_packed, hipEventDisableTiming));
            EMDEE_HIP_CHECK(hipEventCreateWithFlags(&d->ev_done, hipEventDisableTiming));
            EMDEE_HIP_CHECK(hipEventCreateWithFlags(&d->ev_bnd, hipEventDisableTiming));
            const int32_t per[3] = {d->geo.periodic[0], d->geo.periodic[1], d->geo.periodic[2]};
            d->md = std::make_unique<MdImpl<real>>(d->ctx, d->geo.local_lo, d->geo.local_len, per, model, skin);
            d->md->sys.lgv_by_tag = true;                  // the thermostat's noise is keyed by the global ids that travel with the atoms
            if (d->owns_ctx) d->md->caller_ctx = c;        // queries of this engine order their results on the caller's stream
            d->words.ensure(DD_WORDS);
            d->small.ensure(96);
            d->red.ensure(8);
            EMDEE_HIP_CHECK(hipMemsetAsync(d->words.ptr, 0, DD_WORDS * sizeof(int), d->stream()));
        }
        if (use_rccl) {
            RcclApi &api = RcclApi::get();
            api.load();
            RcclApi::UniqueId id;
            memcpy(&id, unique_id, sizeof(id));
            EMDEE_RCCL_CHECK(api.CommInitRank(&comm, world, id, rank_first));
        }
    }

    ~DdImpl() override { release(); }

    // everything the constructor creates, in whatever state it got to (handles that were never created are null)
    void release() {
        (void)hipSetDevice(user_ctx->device);
        for (auto &d : dom) {
            if (d->ctx && d->ctx->stream) (void)hipStreamSynchronize(d->stream());
            if (d->comm) (void)hipStreamSynchronize(d->comm);
            if (d->side) (void)hipStreamSynchronize(d->side);
        }
        if (comm) (void)RcclApi::get().CommDestroy(comm);
        comm = nullptr;
        for (auto &d : dom) {
            d->md.reset();
            if (d->ev_packed) (void)hipEventDestroy(d->ev_packed);
            if (d->ev_done) (void)hipEventDestroy(d->ev_done);
            if (d->ev_bnd) (void)hipEventDestroy(d->ev_bnd);
            if (d->comm) (void)hipStreamDestroy(d->comm);
            if (d->side) (void)hipStreamDestroy(d->side);
            if (d->owns_ctx && d->ctx) {
                if (d->ctx->stream) (void)hipStreamDestroy(d->ctx->stream);
                if (d->ctx->host_flags) (void)hipHostFree(d->ctx->host_flags);
                delete d->ctx;
            }
        }
        dom.clear();
    }

    Domain<real> &local(int l) {
        EMDEE_REQUIRE(l >= 0 && l < n_local, EMDEE_ERR_INVALID, "emdee_dd: local domain %d out of range [0, %d)", l, n_local);
        return *dom[l];
    }
    // ---------------------------------------------------------------- transports
    // Every local domain has filled its Xfer and recorded ev_packed on its compute stream once the send buffer is
    // complete.  Afterwards ev_done (communication stream) marks the arrival of all its messages.
    // Without overlap (emdee_dd_set_overlap 0), a process that holds ONE domain runs every exchange IN ORDER on its compute
    // stream: no event is recorded or waited for (each record / cross-stream wait pair leaves ~25 us of empty queue per
    // step).  Several domains in one process copy from each other's buffers and keep the events.
    // The exchanges of a rebuild (counts, migrants, ghost rows) have nothing to overlap with and always go in order.
    bool in_rebuild = false;
    bool force_inline = false;            // (lock-step halves: the exchange is queued on whatever stream the domain points at)
    bool inline_exchange() const { return (!overlap || in_rebuild || force_inline) && dom.size() == 1 && (use_rccl || mirror || dom[0]->geo.npeers == 0); }
    void record_packed(Domain<real> &d) {
        if (!inline_exchange()) EMDEE_HIP_CHECK(hipEventRecord(d.ev_packed, d.stream()));
    }
    void exchange() {
        const bool inl = inline_exchange();
        if (use_rccl) {
            RcclApi &api = RcclApi::get();
            Domain<real> &d = *dom[0];
            hipStream_t cs = inl ? d.stream() : d.comm;
            if (!inl) EMDEE_HIP_CHECK(hipStreamWaitEvent(d.comm, d.ev_packed, 0));
            if (d.geo.npeers > 0) {
                EMDEE_RCCL_CHECK(api.GroupStart());
                for (int p = 0; p < d.geo.npeers; p++) {
                    if (d.xf.sbytes[p]) EMDEE_RCCL_CHECK(api.Send(d.xf.send + d.xf.soff[p], d.xf.sbytes[p], RcclApi::kChar, d.geo.peers[p], comm, cs));
                    if (d.xf.rbytes[p]) EMDEE_RCCL_CHECK(api.Recv(d.xf.recv + d.xf.roff[p], d.xf.rbytes[p], RcclApi::kChar, d.geo.peers[p], comm, cs));
                }
                EMDEE_RCCL_CHECK(api.GroupEnd());
            }
            if (!inl) EMDEE_HIP_CHECK(hipEventRecord(d.ev_done, d.comm));
            return;
        }
        if (mirror) {
            // every peer is my own image: its message is mine (the pack kernels wrote it as the peer would have)
            Domain<real> &d = *dom[0];
            hipStream_t cs = inl ? d.stream() : d.comm;
            if (!inl) EMDEE_HIP_CHECK(hipStreamWaitEvent(d.comm, d.ev_packed, 0));
            PullArgs pa{};
            unsigned most = 0;
            for (int p = 0; p < d.geo.npeers; p++) {
                EMDEE_REQUIRE(d.xf.sbytes[p] == d.xf.rbytes[p], EMDEE_ERR_STATE, "emdee_dd: replica messages of different sizes (%zu sent, %zu expected)", d.xf.sbytes[p], d.xf.rbytes[p]);
                if (d.xf.rbytes[p] == 0) continue;
                PullSeg &g = pa.seg[pa.n++];
                g.src = reinterpret_cast<const unsigned *>(d.xf.send + d.xf.soff[p]);
                g.dst = reinterpret_cast<unsigned *>(d.xf.recv + d.xf.roff[p]);
                g.words = (unsigned)(d.xf.rbytes[p] / 4);
                most = std::max(most, g.words);
            }
            if (pa.n > 0) hipLaunchKernelGGL(k_dd_pull, dim3(std::min(64u, blocks_for(most, 1024)), pa.n), dim3(256), 0, cs, pa);
            if (!inl) EMDEE_HIP_CHECK(hipEventRecord(d.ev_done, cs));
            return;
        }
        if (inl) return;                                       // one domain without peers: nobody to copy from
        // all domains in this process: ONE communication stream (the first domain's) waits for every domain's send buffer,
        // then every receiver pulls its messages with one launch (k_dd_pull); every domain's ev_done marks the end of ALL
        // pulls, so a domain that has waited for its own may unpack AND pack again (nobody still reads its send buffer)
        hipStream_t cs = dom[0]->comm;
        for (auto &pd : dom) EMDEE_HIP_CHECK(hipStreamWaitEvent(cs, pd->ev_packed, 0));
        for (auto &pd : dom) {
            Domain<real> &d = *pd;
            PullArgs pa{};
            unsigned most = 0;
            for (int p = 0; p < d.geo.npeers; p++) {
                Domain<real> &s = *dom[d.geo.peers[p] - rank_first];
                const int q = s.geo.peer_index(d.geo.rank);
                EMDEE_REQUIRE(q >= 0 && s.xf.sbytes[q] == d.xf.rbytes[p], EMDEE_ERR_STATE,
                              "emdee_dd: message sizes of domains %d and %d disagree (%zu sent, %zu expected)", s.geo.rank,
                              d.geo.rank, q >= 0 ? s.xf.sbytes[q] : (size_t)0, d.xf.rbytes[p]);
                if (d.xf.rbytes[p] == 0) continue;
                EMDEE_REQUIRE(d.xf.rbytes[p] % 4 == 0 && d.xf.roff[p] % 4 == 0 && s.xf.soff[q] % 4 == 0, EMDEE_ERR_STATE,
                              "emdee_dd: a message that is not a whole number of words");
                PullSeg &g = pa.seg[pa.n++];
                g.src = reinterpret_cast<const unsigned *>(s.xf.send + s.xf.soff[q]);
                g.dst = reinterpret_cast<unsigned *>(d.xf.recv + d.xf.roff[p]);
                g.words = (unsigned)(d.xf.rbytes[p] / 4);
                most = std::max(most, g.words);
            }
            if (pa.n > 0)
                hipLaunchKernelGGL(k_dd_pull, dim3(std::min(64u, blocks_for(most, 1024)), pa.n), dim3(256), 0, cs, pa);
        }
        for (auto &pd : dom) EMDEE_HIP_CHECK(hipEventRecord(pd->ev_done, cs));
    }
    // compute streams wait for the arrival of their messages -- and, with in-process copies, for everybody
    // who reads this domain's send buffer, before it is packed again
    void wait_exchange() {
        if (inline_exchange()) return;
        // (in-process copies: a domain's ev_done lies behind the pulls of ALL domains, those that read its send buffer included)
        for (auto &pd : dom) EMDEE_HIP_CHECK(hipStreamWaitEvent(pd->stream(), pd->ev_done, 0));
    }
    // sum of n doubles over all domains (vals: per local domain n values on the host; result in out)
    void allreduce_sum(const std::vector<std::vector<double>> &vals, int n, double *out) {
        for (int k = 0; k < n; k++) out[k] = 0.0;
        for (auto &v : vals)
            for (int k = 0; k < n; k++) out[k] += v[k];
        if (mirror)
            for (int k = 0; k < n; k++) out[k] *= (double)world;   // (every rank of the rehearsed grid holds the same)
        if (!use_rccl) return;
        Domain<real> &d = *dom[0];
        EMDEE_REQUIRE(n <= 8, EMDEE_ERR_INVALID, "allreduce_sum: at most 8 values");
        EMDEE_HIP_CHECK(hipMemcpyAsync(d.red.ptr, out, n * sizeof(double), hipMemcpyHostToDevice, d.stream()));
        EMDEE_RCCL_CHECK(RcclApi::get().AllReduce(d.red.ptr, d.red.ptr, (size_t)n, RcclApi::kFloat64, RcclApi::kSum, comm, d.stream()));
        EMDEE_HIP_CHECK(hipMemcpyAsync(out, d.red.ptr, n * sizeof(double), hipMemcpyDeviceToHost, d.stream()));
        EMDEE_HIP_CHECK(hipStreamSynchronize(d.stream()));
    }

    // ---------------------------------------------------------------- state in
    void set_atoms(int l, int n, const void *pos, const void *vel, const emdee_lj_atom *atoms, const int64_t *gids) override {
        use_device(user_ctx);
        Domain<real> &d = local(l);
        EMDEE_REQUIRE(n >= 0 && (n == 0 || (pos && vel && atoms && gids)), EMDEE_ERR_INVALID, "emdee_dd_set_atoms: NULL array");
        d.x.ensure(3 * (size_t)n + 3); d.v.ensure(3 * (size_t)n + 3); d.at.ensure((size_t)n + 1); d.gid.ensure((size_t)n + 1);
        hipStream_t s = user_ctx->stream;
        if (n > 0) {
            EMDEE_HIP_CHECK(hipMemcpyAsync(d.x.ptr, pos, 3 * (size_t)n * sizeof(real), hipMemcpyDeviceToDevice, s));
            EMDEE_HIP_CHECK(hipMemcpyAsync(d.v.ptr, vel, 3 * (size_t)n * sizeof(real), hipMemcpyDeviceToDevice, s));
            EMDEE_HIP_CHECK(hipMemcpyAsync(d.at.ptr, atoms, (size_t)n * sizeof(emdee_lj_atom), hipMemcpyDeviceToDevice, s));
            EMDEE_HIP_CHECK(hipMemcpyAsync(d.gid.ptr, gids, (size_t)n * sizeof(long long), hipMemcpyDeviceToDevice, s));
        }
        EMDEE_HIP_CHECK(hipStreamSynchronize(s));
        d.n_owned = n;
        d.n_ghost = 0;
        loaded = false;
    }

    // collective: hand every atom to the brick that contains it, choose and exchange the ghosts, load the engines
    void load() override {
        use_device(user_ctx);
        for (auto &d : dom) d->sys().uniform_known = -1;
        redistribute(false);
        agree_on_species();
        if (n_global == 0) {
            std::vector<std::vector<double>> v;
            for (auto &d : dom) v.push_back({(double)d->n_owned});
            double tot = 0;
            allreduce_sum(v, 1, &tot);
            n_global = (int64_t)(tot + 0.5);
        }
        loaded = true;
    }

    // stable partition of items 0..n-1 by mask bits into nbins bins: counts and scanned offsets (device), starts and
    // per-peer counts into small[0..] / small[33..]
    // One species everywhere, or not?  Every engine looked at its own atoms (owned + ghosts) during the load; agree once
    // over all domains, so that later loads -- atoms only change owner -- skip that scan and its read-back.
    void agree_on_species() {
        // (a domain without atoms has no opinion -- and takes the others' constants, for the atoms it may receive later)
        auto bits = [](float f) { uint32_t u; memcpy(&u, &f, 4); return (double)u; };
        auto unbits = [](double b) { uint32_t u = (uint32_t)(b + 0.5); float f; memcpy(&f, &u, 4); return f; };
        std::vector<std::vector<double>> v;
        for (auto &d : dom) {
            const bool has = d->sys().n_total > 0;
            v.push_back({has ? 1.0 : 0.0, has && d->sys().uniform_atoms ? 1.0 : 0.0, has ? bits(d->sys().uni_first.half_sigma) : 0.0,
                         has ? bits(d->sys().uni_first.twice_sqrt_eps) : 0.0});
        }
        double sum[4];
        allreduce_sum(v, 4, sum);
        const double voters = sum[0];
        std::vector<std::vector<double>> ok;
        for (auto &d : dom) {
            const bool has = d->sys().n_total > 0;
            const bool same = voters > 0 && sum[1] == voters &&
                              (!has || (sum[2] == voters * bits(d->sys().uni_first.half_sigma) && sum[3] == voters * bits(d->sys().uni_first.twice_sqrt_eps)));
            ok.push_back({same ? 1.0 : 0.0});
        }
        double agreed = 0;
        allreduce_sum(ok, 1, &agreed);
        const bool uni = agreed == (double)world;
        for (auto &d : dom) {
            d->sys().uniform_known = uni ? 1 : 0;
            d->sys().species_known.n = 1;
            if (uni) d->sys().set_uniform_constants(emdee_lj_atom{unbits(sum[2] / voters), unbits(sum[3] / voters)});
        }
        if (uni || voters == 0) return;
        // Two species everywhere?  Then every engine sorts by (cell, species) and takes the typed kernels where they pay
        // (typed.hpp).  Every domain with atoms must have found the same two LJAtom bit patterns (an empty one has no opinion).
        auto half = [](unsigned long long k, int h) { return (double)(unsigned)(h ? (k >> 32) : (k & 0xffffffffull)); };
        std::vector<std::vector<double>> t;
        for (auto &d : dom) {
            const NbSystem<real> &e = d->sys();
            const bool two = e.n_total > 0 && e.species.n == 2;
            t.push_back({e.n_total > 0 ? 1.0 : 0.0, two ? 1.0 : 0.0, two ? half(e.species.key[0], 0) : 0.0, two ? half(e.species.key[0], 1) : 0.0,
                         two ? half(e.species.key[1], 0) : 0.0, two ? half(e.species.key[1], 1) : 0.0});
        }
        double ts[6];
        allreduce_sum(t, 6, ts);
        std::vector<std::vector<double>> ok2;
        for (auto &d : dom) {
            const NbSystem<real> &e = d->sys();
            bool same = ts[0] > 0 && ts[1] == ts[0];
            if (same && e.n_total > 0)
                same = ts[2] == ts[0] * half(e.species.key[0], 0) && ts[3] == ts[0] * half(e.species.key[0], 1) &&
                       ts[4] == ts[0] * half(e.species.key[1], 0) && ts[5] == ts[0] * half(e.species.key[1], 1);
            ok2.push_back({same ? 1.0 : 0.0});
        }
        double agreed2 = 0;
        allreduce_sum(ok2, 1, &agreed2);
        if (agreed2 != (double)world) return;
        SpeciesTable tab{2, {0, 0, 0, 0}};
        for (int k = 0; k < 2; k++)
            tab.key[k] = (unsigned long long)(unsigned)(ts[2 + 2 * k] / ts[0] + 0.5) | ((unsigned long long)(unsigned)(ts[3 + 2 * k] / ts[0] + 0.5) << 32);
        for (auto &d : dom) d->sys().species_known = tab;
    }

    // stable partition of items 0..n-1 by mask bits into nbins bins.  partition_prepare sizes and clears the per-block
    // counts, the kernel that produces the masks fills them (part_count_block), partition_finish scans them and leaves the
    // bin starts in small[0..] and the per-peer counts in small[33..]
    int partition_prepare(Domain<real> &d, int n, int nbins) {
        const int nblocks = std::max(1, (int)blocks_for(n, PART_BLOCK));
        d.counts.ensure((size_t)nbins * nblocks + 2);
        EMDEE_HIP_CHECK(hipMemsetAsync(d.counts.ptr, 0, ((size_t)nbins * nblocks + 1) * sizeof(int), d.stream()));
        return nblocks;
    }
    void partition_finish(Domain<real> &d, int nbins, int nblocks, const DdBins &pb) {
        d.scanner.run(d.counts.ptr, (size_t)nbins * nblocks + 1, d.stream());
        hipLaunchKernelGGL(k_part_starts, dim3(1), dim3(64), 0, d.stream(), nbins, nblocks, d.counts.ptr, d.small.ptr, pb, d.small.ptr + 33);
    }
    void partition_scatter(Domain<real> &d, int n, int nbins, int total, bool with_bins) {
        const int nblocks = std::max(1, (int)blocks_for(n, PART_BLOCK));
        d.ids.ensure((size_t)total + 1);
        if (with_bins) d.bins.ensure((size_t)total + 1);
        if (n > 0)
            hipLaunchKernelGGL(k_part_scatter, dim3(nblocks), dim3(PART_BLOCK), 0, d.stream(), n, d.mask.ptr, nbins, nblocks,
                               d.counts.ptr, d.ids.ptr, with_bins ? d.bins.ptr : nullptr);
    }
    // exchange the per-peer counts in small[33..33+npeers) -> small[60..60+npeers), then read small back (blocking)
    void exchange_counts_and_read() {
        for (auto &pd : dom) {
            Domain<real> &d = *pd;
            d.xf.send = reinterpret_cast<const unsigned char *>(d.small.ptr + 33);
            d.xf.recv = reinterpret_cast<unsigned char *>(d.small.ptr + 60);
            for (int p = 0; p < d.geo.npeers; p++) {
                d.xf.soff[p] = d.xf.roff[p] = (size_t)p * sizeof(int);
                d.xf.sbytes[p] = d.xf.rbytes[p] = sizeof(int);
            }
            record_packed(d);
        }
        exchange();
        wait_exchange();
        if (dom.size() == 1) {
            read_back_words(dom[0]->ctx, dom[0]->stream(), dom[0]->small.ptr, 96, dom[0]->host_small);
            return;
        }
        for (auto &pd : dom)
            EMDEE_HIP_CHECK(hipMemcpyAsync(pd->host_small, pd->small.ptr, 96 * sizeof(int), hipMemcpyDeviceToHost, pd->stream()));
        for (auto &pd : dom) EMDEE_HIP_CHECK(hipStreamSynchronize(pd->stream()));
    }
    // variable-size rows: send counts small[33+p], receive counts small[60+p] (host copies), row_bytes each
    void exchange_rows(size_t row_bytes) {
        for (auto &pd : dom) {
            Domain<real> &d = *pd;
            size_t so = 0, ro = 0;
            for (int p = 0; p < d.geo.npeers; p++) {
                d.xf.soff[p] = so; d.xf.sbytes[p] = (size_t)d.host_small[33 + p] * row_bytes; so += d.xf.sbytes[p];
                d.xf.roff[p] = ro; d.xf.rbytes[p] = (size_t)d.host_small[60 + p] * row_bytes; ro += d.xf.rbytes[p];
            }
            d.xf.send = d.sendbuf.ptr;
            d.xf.recv = d.recvbuf.ptr;
            record_packed(d);
        }
        exchange();
        wait_exchange();
    }

    // The engine's state as dense caller-order arrays (owned atoms first, then the ghosts; positions as the records hold them):
    // what the rebuild WITH counts starts from -- the first rebuild after a load, and the redo after a message overflowed.
    // `ids`, optional: engine ids (a send list) to translate into positions of those arrays.
    void export_caller_arrays(Domain<real> &d, int *ids_to_map = nullptr, int n_ids = 0) {
        const size_t nt = (size_t)d.n_owned + d.n_ghost;
        d.x.ensure(3 * nt + 3); d.v.ensure(3 * (size_t)d.n_owned + 3); d.at.ensure(nt + 1); d.gid.ensure(nt + 1);
        EMDEE_REQUIRE(d.sys().n_total == (int)nt, EMDEE_ERR_STATE, "emdee_dd: domain %d holds %d atoms, its engine %d", d.geo.rank, (int)nt, d.sys().n_total);
        d.sys().unsort(d.x.ptr, d.v.ptr, nullptr, nullptr, nullptr, d.at.ptr, d.gid.ptr, true);
        if (ids_to_map && n_ids > 0) {
            const int *map = d.sys().ids_map();
            if (map) hipLaunchKernelGGL(k_dd_map_ids, dim3(blocks_for(n_ids, 256)), dim3(256), 0, d.stream(), n_ids, map, ids_to_map);
        }
    }

    // with_forces = false: the caller follows up with step_after_rebuild, whose fused kernel evaluates them
    void redistribute(bool from_engines, bool with_forces = true) {
        struct Scope {
            bool &f;
            explicit Scope(bool &b) : f(b) { f = true; }
            ~Scope() { f = false; }
        } scope(in_rebuild);
        struct Wall {
            double &ms;
            int64_t &n;
            std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
            ~Wall() { ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); n++; }
        } wall{stat_rebuild_wall_ms, stat_rebuild_calls};
        struct Reads {                                             // blocking read-backs inside rebuilds (all local domains' contexts)
            DdImpl *dd;
            int64_t at;
            static int64_t now(DdImpl *d) {
                int64_t n = 0;
                std::vector<const emdee_ctx *> seen;
                for (auto &pd : d->dom) {
                    if (std::find(seen.begin(), seen.end(), pd->ctx) != seen.end()) continue;
                    seen.push_back(pd->ctx);
                    n += pd->ctx->readbacks;
                }
                return n;
            }
            explicit Reads(DdImpl *d) : dd(d), at(now(d)) {}
            ~Reads() { dd->stat_rebuild_readbacks += now(dd) - at; }
        } reads(this);
        join_halo();
        for (auto &pd : dom) pd->words_clear = false;
        if (world == 1 && from_engines && !no_shortcut) {
            // one domain, no cut: nobody to hand atoms to and no ghosts -- the engine's own re-sort (same list, same forces,
            // none of the ownership passes)
            Domain<real> &d = *dom[0];
            d.md->rebuild();
            if (with_forces) d.md->forces(EMDEE_FORCES, 0);
            d.since_build = 0;
            EMDEE_HIP_CHECK(hipMemsetAsync(d.words.ptr, 0, DD_WORDS * sizeof(int), d.stream()));
            stat_rebuilds++;
            return;
        }
        // The count-free protocol rests on every rank neighbouring every other one: the overflow word and the error word of a
        // rank reach everybody in the SAME exchange (message headers), so all commit or all redo.  Today that follows from
        // the <= 3 bricks per dimension of DdGeom::init; should that limit ever go, a far rank would commit while its
        // neighbours redo -- so the dependency is checked here, and such a grid takes the counted rebuild.
        bool all_neighbours = true;
        for (auto &pd : dom) all_neighbours = all_neighbours && pd->geo.npeers == world - 1;
        if (from_engines && count_free && all_neighbours && dom[0]->have_caps) {
            if (redistribute_sorted(with_forces)) return;
            stat_fallback++;                                        // a capacity was exceeded somewhere: everybody redoes it with counts
        }
        // ---- 0. caller-order copies of the integrated state
        for (auto &pd : dom) {
            Domain<real> &d = *pd;
            if (from_engines) export_caller_arrays(d);
            d.mask.ensure((size_t)d.n_owned + 1);
            EMDEE_HIP_CHECK(hipMemsetAsync(d.small.ptr, 0, 96 * sizeof(int), d.stream()));
            // ---- 1. owner of every atom; partition into stay | one bin per peer
            const int nb1 = 1 + d.geo.npeers, nblk1 = partition_prepare(d, d.n_owned, nb1);
            if (d.n_owned > 0)
                hipLaunchKernelGGL((k_dd_classify<real>), dim3(nblk1), dim3(PART_BLOCK), 0, d.stream(), d.n_owned, d.x.ptr,
                                   d.geo.template device<real>(), d.mask.ptr, d.small.ptr + 95, nb1, nblk1, d.counts.ptr);
            DdBins pb{};
            pb.npeers = d.geo.npeers;
            for (int p = 0; p <= d.geo.npeers + 1; p++) pb.lo[p] = std::min(1 + p, 1 + d.geo.npeers);
            partition_finish(d, nb1, nblk1, pb);
            // "an atom left the neighbourhood of its brick" travels with the counts: every rank fails in the same REQUIRE below,
            // instead of one throwing while its peers walk into the exchange of rows with a rank that is gone
            hipLaunchKernelGGL(k_dd_flag_counts, dim3(1), dim3(64), 0, d.stream(), d.geo.npeers, d.small.ptr + 95, d.small.ptr + 33);
        }
        exchange_counts_and_read();
        // ---- 2. move the leavers
        for (auto &pd : dom) {
            Domain<real> &d = *pd;
            bool err = d.host_small[95] != 0;
            for (int p = 0; p < d.geo.npeers; p++) {
                err = err || (d.host_small[60 + p] & DD_COUNT_ERR) != 0;
                d.host_small[33 + p] &= ~DD_COUNT_ERR; d.host_small[60 + p] &= ~DD_COUNT_ERR;
            }
            EMDEE_REQUIRE(!err, EMDEE_ERR_STATE, "emdee_dd: an atom left the neighbourhood of its brick (seen by domain %d: its own atom or a peer's -- the word travels with the counts, every rank fails here together)", d.geo.rank);
            const int nb = 1 + d.geo.npeers, total = d.host_small[nb], n_stay = d.host_small[1];
            partition_scatter(d, d.n_owned, nb, total, false);
            const int n_leave = total - n_stay;
            int n_arrive = 0;
            for (int p = 0; p < d.geo.npeers; p++) n_arrive += d.host_small[60 + p];
            d.sendbuf.ensure((size_t)n_leave * sizeof(MigRow<real>) + 64);
            d.recvbuf.ensure((size_t)n_arrive * sizeof(MigRow<real>) + 64);
            if (n_leave > 0) {
                DdBins st{};
                st.npeers = d.geo.npeers;
                for (int b = 0; b <= d.geo.npeers + 1 && b < DD_MAX_PEERS + 2; b++) st.lo[b] = d.host_small[std::min(b, nb)];
                hipLaunchKernelGGL((k_dd_pack_migrants<real>), dim3(blocks_for(n_leave, 256)), dim3(256), 0, d.stream(), n_stay, total,
                                   d.ids.ptr, d.x.ptr, d.v.ptr, d.at.ptr, d.gid.ptr, reinterpret_cast<MigRow<real> *>(d.sendbuf.ptr),
                                   d.geo.template device<real>(), st);
            }
            stat_migrated += n_leave;
        }
        exchange_rows(sizeof(MigRow<real>));
        for (auto &pd : dom) {
            Domain<real> &d = *pd;
            const int n_stay = d.host_small[1];
            int n_arrive = 0;
            for (int p = 0; p < d.geo.npeers; p++) n_arrive += d.host_small[60 + p];
            const int n_new = n_stay + n_arrive;
            // (ghost capacity is added below, once the ghost count is known; x2 holds the owned part for now)
            d.x2.ensure(3 * (size_t)n_new + 3); d.v2.ensure(3 * (size_t)n_new + 3); d.at2.ensure((size_t)n_new + 1); d.gid2.ensure((size_t)n_new + 1);
            if (n_new > 0)
                hipLaunchKernelGGL((k_dd_assemble_owned<real>), dim3(blocks_for(n_new, 256)), dim3(256), 0, d.stream(), n_stay, n_arrive,
                                   d.ids.ptr, d.x.ptr, d.v.ptr, d.at.ptr, d.gid.ptr, reinterpret_cast<const MigRow<real> *>(d.recvbuf.ptr),
                                   d.x2.ptr, d.v2.ptr, d.at2.ptr, d.gid2.ptr);
            d.x.swap(d.x2); d.v.swap(d.v2); d.at.swap(d.at2); d.gid.swap(d.gid2);
            d.n_owned = n_new;
            // ---- 3. ghosts: which neighbours need which of my atoms
            d.mask.ensure((size_t)n_new + 1);
            EMDEE_HIP_CHECK(hipMemsetAsync(d.small.ptr, 0, 96 * sizeof(int), d.stream()));
            const int n_sel = d.geo.ghost_nbins > 0 ? n_new : 0, nb2 = std::max(1, d.geo.ghost_nbins);
            const int nblk2 = partition_prepare(d, n_sel, nb2);
            if (n_sel > 0)
                hipLaunchKernelGGL((k_dd_ghost_mask<real>), dim3(nblk2), dim3(PART_BLOCK), 0, d.stream(), n_sel, d.x.ptr,
                                   d.geo.template device<real>(), d.mask.ptr, nb2, nblk2, d.counts.ptr);
            DdBins pb{};
            pb.npeers = d.geo.npeers;
            for (int p = 0; p <= d.geo.npeers + 1; p++) pb.lo[p] = d.geo.peer_bin_lo[std::min(p, d.geo.npeers)];
            partition_finish(d, nb2, nblk2, pb);
        }
        exchange_counts_and_read();
        for (auto &pd : dom) {
            Domain<real> &d = *pd;
            const int nb = std::max(1, d.geo.ghost_nbins), total = d.geo.ghost_nbins > 0 ? d.host_small[nb] : 0;
            partition_scatter(d, d.geo.ghost_nbins > 0 ? d.n_owned : 0, nb, total, true);
            d.n_send = total;
            d.n_ghost = 0;
            d.plan = DdPlan{};
            d.plan.npeers = d.geo.npeers;
            for (int p = 0; p < d.geo.npeers; p++) {
                d.plan.send_start[p + 1] = d.plan.send_start[p] + d.host_small[33 + p];
                d.plan.recv_start[p + 1] = d.plan.recv_start[p] + d.host_small[60 + p];
            }
            d.n_ghost = d.plan.recv_start[d.geo.npeers];
            EMDEE_REQUIRE(d.plan.send_start[d.geo.npeers] == total, EMDEE_ERR_STATE, "emdee_dd: ghost send list inconsistent");
            d.codes.ensure((size_t)total + 1);
            d.sendbuf.ensure((size_t)total * sizeof(GhostRow<real>) + 64);
            d.recvbuf.ensure((size_t)d.n_ghost * sizeof(GhostRow<real>) + 64);
            // room for the ghosts behind the owned atoms (contents preserved by hand: DevBuf::ensure does not)
            grow_keep(d.x, 3 * (size_t)d.n_owned, 3 * (size_t)(d.n_owned + d.n_ghost) + 3, d.stream());
            grow_keep(d.at, (size_t)d.n_owned, (size_t)(d.n_owned + d.n_ghost) + 1, d.stream());
            grow_keep(d.gid, (size_t)d.n_owned, (size_t)(d.n_owned + d.n_ghost) + 1, d.stream());
            if (total > 0)
                hipLaunchKernelGGL((k_dd_pack_ghost_rows<real>), dim3(blocks_for(total, 256)), dim3(256), 0, d.stream(), total,
                                   d.ids.ptr, d.bins.ptr, d.geo.template device<real>(), d.x.ptr, d.at.ptr, d.gid.ptr,
                                   reinterpret_cast<GhostRow<real> *>(d.sendbuf.ptr), d.codes.ptr);
        }
        exchange_rows(sizeof(GhostRow<real>));
        // ---- 4. load the engines: bin, sort, neighbour list, forces
        for (auto &pd : dom) {
            Domain<real> &d = *pd;
            if (d.n_ghost > 0)
                hipLaunchKernelGGL((k_dd_unpack_ghost_rows<real>), dim3(blocks_for(d.n_ghost, 256)), dim3(256), 0, d.stream(), d.n_ghost,
                                   reinterpret_cast<const GhostRow<real> *>(d.recvbuf.ptr), d.x.ptr + 3 * (size_t)d.n_owned,
                                   d.at.ptr + d.n_owned, d.gid.ptr + d.n_owned);
            load_engine(d, with_forces);
        }
        stat_rebuilds++;
    }

    // capacity of a ghost message at the next rebuild, from its rows at this one (both ends hold the same number)
    int ghost_cap(int rows) const { return ghost_cap_exact ? rows : rows + rows / 8 + 64; }
    void set_caps(Domain<real> &d) {
        const int np = d.geo.npeers;
        const int64_t per_rank = n_global > 0 ? n_global / world : (int64_t)d.n_owned;
        int mig = (int)std::max<int64_t>(256, per_rank / 512);
        if (mig_cap_forced > 0) mig = mig_cap_forced;
        d.mig_caps.npeers = d.gs_caps.npeers = d.gr_caps.npeers = np;
        d.mig_caps.start[0] = d.gs_caps.start[0] = d.gr_caps.start[0] = 0;
        d.mig_caps.debug_inject = d.gs_caps.debug_inject = d.gr_caps.debug_inject = 0;
#ifdef EMDEE_BOUNDS
        // the checker's own test: pack ghost rows with the send list indexed by the counts although a message overflowed
        // (what the padded ghost pack did before 2d85cf1 -- an out-of-range index the bounds build must report, not use)
        if (const char *e = std::getenv("EMDEE_BOUNDS_INJECT")) d.gs_caps.debug_inject = std::string(e) == "ghost_pack" ? 1 : 0;
#endif
        for (int p = 0; p < np; p++) {
            d.mig_caps.start[p + 1] = d.mig_caps.start[p] + mig;
            d.gs_caps.start[p + 1] = d.gs_caps.start[p] + ghost_cap(d.plan.send_start[p + 1] - d.plan.send_start[p]);
            d.gr_caps.start[p + 1] = d.gr_caps.start[p] + ghost_cap(d.plan.recv_start[p + 1] - d.plan.recv_start[p]);
        }
        d.have_caps = n_global > 0;       // (the first load runs before the global count is known)
    }
    // slots an engine needs for its next rebuild in its own order: the state, the arrivals' and the ghosts' capacities behind it
    size_t edit_slots(const Domain<real> &d, size_t n_now) const {
        const int np = d.geo.npeers;
        return (n_now + PART_BLOCK) / PART_BLOCK * PART_BLOCK + PART_BLOCK + (size_t)d.mig_caps.start[np] + (size_t)d.gr_caps.start[np];
    }
    // step 4 of a rebuild with counts: bin, sort, neighbour list (forces) from the caller-order arrays
    void load_engine(Domain<real> &d, bool with_forces) {
        set_caps(d);
        // (room for the rebuilds to come: atoms drift in and out, the ghost capacities follow the ghost counts)
        const size_t n_now = (size_t)d.n_owned + d.n_ghost;
        d.sys().cap_hint = d.have_caps ? edit_slots(d, n_now + n_now / 16 + 1024) + (size_t)d.gr_caps.start[d.geo.npeers] / 8 : 0;
        d.md->tags_user = reinterpret_cast<const long long *>(d.gid.ptr);
        d.md->defer_forces = !with_forces;
        d.md->set_state(d.n_owned, d.n_ghost, d.x.ptr, d.v.ptr, d.at.ptr, nullptr);
        d.md->defer_forces = false;
        d.md->tags_user = nullptr;
        for (int p = 0; p <= d.geo.npeers; p++) d.plan.ghost_id[p] = d.n_owned + d.plan.recv_start[p];
        finish_rebuild(d);
    }
    // buffers of the per-step messages (the padded messages of the next rebuild share them), the guard words
    void finish_rebuild(Domain<real> &d) {
        d.since_build = 0;
        // per-step messages: header + 3 reals per atom and peer
        const size_t w = sizeof(real);
        const int np = d.geo.npeers;
        const size_t pad_s = std::max(dd_pad_total(d.mig_caps, sizeof(MigRow<real>)), dd_pad_total(d.gs_caps, sizeof(GhostRow<real>)));
        const size_t pad_r = std::max(dd_pad_total(d.mig_caps, sizeof(MigRow<real>)), dd_pad_total(d.gr_caps, sizeof(GhostRow<real>)));
        d.sendbuf.ensure(std::max(dd_msg_begin(d.plan.send_start, np, w), pad_s) + 64);
        d.recvbuf.ensure(std::max(dd_msg_begin(d.plan.recv_start, np, w), pad_r) + 64);
        // (the guard words are cleared by whoever steps next: emdee_dd_step at its start, step_after_rebuild)
    }

    // A rebuild in the engines' own order (dd_kernels.hpp): two padded exchanges, device-side counts throughout, the engine's
    // own re-sort with the leavers struck out and the arrivals and ghosts appended, ONE read-back (with the build's words).
    // False: a message capacity was exceeded on some rank -- every rank sees that in the headers it received, every engine
    // has rolled back to the state it had, and the caller redoes the rebuild with exact counts.
    bool redistribute_sorted(bool with_forces) {
        const size_t mrow = sizeof(MigRow<real>), grow = sizeof(GhostRow<real>);
        struct Slots { int n, q_arr, mig, gr, ghost_base, n_items, nblk1, nblk_arr, nblk_g; };
        std::vector<Slots> S(dom.size());
        for (size_t l = 0; l < dom.size(); l++) {
            Domain<real> &d = *dom[l];
            const int np = d.geo.npeers;
            Slots &e = S[l];
            e.n = d.sys().n_total;
            e.mig = d.mig_caps.start[np]; e.gr = d.gr_caps.start[np];
            e.q_arr = std::max(1, (e.n + PART_BLOCK - 1) / PART_BLOCK) * PART_BLOCK;
            e.nblk1 = e.q_arr / PART_BLOCK;
            e.nblk_arr = std::max(1, (e.mig + PART_BLOCK - 1) / PART_BLOCK);
            e.nblk_g = e.nblk1 + e.nblk_arr;
            e.ghost_base = e.q_arr + e.mig;
            e.n_items = e.ghost_base + e.gr;
            if (!d.sys().edit_fits(e.n_items)) {
                // This engine's arrays do not hold the slots (the state grew past the room its last load left, or that load was the
                // very first one): load it again from its own state with room to spare -- a local matter, the peers need not
                // know (same atoms, same order), and the rebuild goes on in the engine's order like everybody's.
                export_caller_arrays(d);
                d.sys().cap_hint = edit_slots(d, (size_t)e.n + e.n / 16 + 1024) + (size_t)e.gr / 8;
                d.md->tags_user = reinterpret_cast<const long long *>(d.gid.ptr);
                d.md->defer_forces = true;
                d.md->set_state(d.n_owned, d.n_ghost, d.x.ptr, d.v.ptr, d.at.ptr, nullptr);
                d.md->defer_forces = false;
                d.md->tags_user = nullptr;
                stat_regrown++;
                EMDEE_REQUIRE(d.sys().edit_fits(e.n_items), EMDEE_ERR_STATE, "emdee_dd: domain %d cannot re-sort %d slots in place", d.geo.rank, e.n_items);
            }
        }
        for (size_t l = 0; l < dom.size(); l++) {
            Domain<real> &d = *dom[l];
            const Slots &e = S[l];
            NbSystem<real> &sy = d.sys();
            const int np = d.geo.npeers, nb1 = 1 + np, nb2 = std::max(1, d.geo.ghost_nbins);
            d.keep.ensure((size_t)e.n_items + 1);
            d.mask.ensure((size_t)e.q_arr + 1);
            d.gmask.ensure((size_t)e.q_arr + (size_t)e.nblk_arr * PART_BLOCK + 1);
            d.w.ensure(DDW_COUNT);
            d.counts.ensure((size_t)nb1 * e.nblk1 + 2);
            d.counts2.ensure((size_t)nb2 * e.nblk_g + 2);
            d.ids.ensure((size_t)std::max(e.mig + PART_BLOCK, d.gs_caps.start[np]) + 1);
            d.bins.ensure((size_t)d.gs_caps.start[np] + 1);
            d.codes.ensure((size_t)d.gs_caps.start[np] + 1);
            // (one launch: the error word, both count arrays, and the guard words of the step that follows the rebuild)
            Zeros().add(d.small.ptr + 95, 1).add(d.counts.ptr, (size_t)nb1 * e.nblk1 + 1).add(d.counts2.ptr, (size_t)nb2 * e.nblk_g + 1)
                .add(d.words.ptr, DD_WORDS).run(d.stream());
            d.words_clear = true;
            // ---- 1. owner of every atom, where the engine keeps it; the ghost directions of those that stay
            hipLaunchKernelGGL((k_dd_classify_sorted<real>), dim3(e.nblk1), dim3(PART_BLOCK), 0, d.stream(), e.n, sy.n_owned, sy.perm.ptr,
                               sy.rec.ptr, d.geo.template device<real>(), d.keep.ptr, d.mask.ptr, d.gmask.ptr, d.small.ptr + 95, nb1,
                               e.nblk1, d.counts.ptr, nb2, e.nblk_g, d.counts2.ptr);
            d.scanner.run(d.counts.ptr, (size_t)nb1 * e.nblk1 + 1, d.stream());
            if (np > 0)
                hipLaunchKernelGGL(k_part_scatter, dim3(e.nblk1), dim3(PART_BLOCK), 0, d.stream(), e.q_arr, d.mask.ptr, nb1, e.nblk1,
                                   d.counts.ptr, d.ids.ptr, (int *)nullptr, (const int *)nullptr, e.mig + PART_BLOCK);
            // ---- 2. the leavers travel in padded messages
            const int nt = std::max(1, std::max(np, e.mig));
            if (np > 0)
                hipLaunchKernelGGL((k_dd_pack_migrants_sorted<real>), dim3(blocks_for(nt, 256)), dim3(256), 0, d.stream(), d.mig_caps,
                                   PartView{d.counts.ptr, e.nblk1}, d.ids.ptr, sy.rec.ptr, sy.te.ptr, sy.vel.ptr, sy.pitch, sy.tag.ptr, d.sendbuf.ptr,
                                   d.geo.template device<real>());
            for (int p = 0; p < np; p++) {
                d.xf.soff[p] = d.xf.roff[p] = dd_pad_msg_begin(d.mig_caps, p, mrow);
                d.xf.sbytes[p] = d.xf.rbytes[p] = dd_pad_msg_bytes(d.mig_caps, p, mrow);
            }
            d.xf.send = d.sendbuf.ptr;
            d.xf.recv = d.recvbuf.ptr;
            record_packed(d);
        }
        exchange();
        wait_exchange();
        for (size_t l = 0; l < dom.size(); l++) {
            Domain<real> &d = *dom[l];
            const Slots &e = S[l];
            NbSystem<real> &sy = d.sys();
            const int np = d.geo.npeers, nb2 = std::max(1, d.geo.ghost_nbins);
            // ---- 3. arrivals behind the old state; ghosts: which neighbours need which of my (new) atoms
            hipLaunchKernelGGL((k_dd_unpack_arrivals<real>), dim3(e.nblk_arr), dim3(PART_BLOCK), 0, d.stream(), d.mig_caps,
                               PartView{d.counts.ptr, e.nblk1}, d.small.ptr + 95, d.recvbuf.ptr, d.n_owned, e.q_arr, sy.rec.ptr, sy.te.ptr, sy.vel.ptr, sy.pitch, sy.tag.ptr,
                               d.keep.ptr, d.gmask.ptr, d.geo.template device<real>(), nb2, e.nblk_g, d.counts2.ptr, e.nblk1, d.w.ptr);
            DdBins pb{};
            pb.npeers = np;
            for (int p = 0; p <= np + 1; p++) pb.lo[p] = d.geo.peer_bin_lo[std::min(p, np)];
            d.scanner.run(d.counts2.ptr, (size_t)nb2 * e.nblk_g + 1, d.stream());
            if (d.geo.ghost_nbins > 0)
                hipLaunchKernelGGL(k_part_scatter, dim3(e.nblk_g), dim3(PART_BLOCK), 0, d.stream(), e.ghost_base, d.gmask.ptr, nb2, e.nblk_g,
                                   d.counts2.ptr, d.ids.ptr, d.bins.ptr, (const int *)nullptr, d.gs_caps.start[np]);
            const int nt = std::max(1, std::max(np, d.gs_caps.start[np]));
            hipLaunchKernelGGL((k_dd_pack_ghost_rows_sorted<real>), dim3(blocks_for(nt, 256)), dim3(256), 0, d.stream(), d.gs_caps,
                               PartView{d.counts2.ptr, e.nblk_g}, pb, d.ids.ptr, d.bins.ptr, d.geo.template device<real>(), sy.rec.ptr, sy.te.ptr, sy.tag.ptr,
                               d.sendbuf.ptr, d.codes.ptr, d.w.ptr);
            for (int p = 0; p < np; p++) {
                d.xf.soff[p] = dd_pad_msg_begin(d.gs_caps, p, grow); d.xf.sbytes[p] = dd_pad_msg_bytes(d.gs_caps, p, grow);
                d.xf.roff[p] = dd_pad_msg_begin(d.gr_caps, p, grow); d.xf.rbytes[p] = dd_pad_msg_bytes(d.gr_caps, p, grow);
            }
            d.xf.send = d.sendbuf.ptr;
            d.xf.recv = d.recvbuf.ptr;
            record_packed(d);
        }
        exchange();
        wait_exchange();
        // ---- 4. the engines re-sort their own slots and build their lists; the counts come back with the build's words
        std::vector<char> built(dom.size(), 0);
        for (size_t l = 0; l < dom.size(); l++) {
            Domain<real> &d = *dom[l];
            const Slots &e = S[l];
            NbSystem<real> &sy = d.sys();
            const int np = d.geo.npeers;
            DdBins pb4{};
            pb4.npeers = np;
            for (int p = 0; p <= np + 1; p++) pb4.lo[p] = d.geo.peer_bin_lo[std::min(p, np)];
            hipLaunchKernelGGL((k_dd_unpack_ghost_rows_sorted<real>), dim3(blocks_for(std::max(1, e.gr), 256)), dim3(256), 0, d.stream(),
                               d.gs_caps, d.gr_caps, PartView{d.counts2.ptr, e.nblk_g}, pb4, d.recvbuf.ptr, e.ghost_base, sy.rec.ptr, sy.te.ptr, sy.vel.ptr,
                               sy.pitch, sy.tag.ptr, d.keep.ptr, d.w.ptr);
            typename NbSystem<real>::EditWords extra;
            extra.dev = d.w.ptr; extra.n = DDW_COUNT; extra.host = d.host_w;
            built[l] = sy.resort_edit(e.n_items, d.keep.ptr, e.ghost_base, np > 0, d.w.ptr + DDW_NLIVE, extra) ? 1 : 0;
        }
        bool over = false;
        for (auto &pd : dom) {
            Domain<real> &d = *pd;
            EMDEE_REQUIRE(d.host_w[DDW_ERR] == 0, EMDEE_ERR_STATE, "emdee_dd: an atom left the neighbourhood of its brick (seen by domain %d: its own atom or a peer's -- the word travels with the ghost messages, every rank fails here together)", d.geo.rank);
            over = over || d.host_w[DDW_OVER] != 0;
        }
        // the same word on every rank (k_dd_unpack_ghost_rows_sorted): ranks in separate processes have nothing but that to agree on
        // the redo, so the validation mode -- all domains here -- insists on it
        for (auto &pd : dom)
            EMDEE_REQUIRE((pd->host_w[DDW_OVER] != 0) == over, EMDEE_ERR_STATE,
                          "emdee_dd: the domains disagree on whether a rebuild message overflowed (domain %d)", pd->geo.rank);
        if (over) {
            for (auto &pd : dom) { pd->sys().rollback_edit(); pd->words_clear = false; }
            return false;
        }
        for (size_t l = 0; l < dom.size(); l++) {
            Domain<real> &d = *dom[l];
            const Slots &e = S[l];
            NbSystem<real> &sy = d.sys();
            const int np = d.geo.npeers;
            EMDEE_REQUIRE(d.host_w[DDW_NLIVE] == d.host_w[DDW_NNEW] + d.host_w[DDW_NGHOST], EMDEE_ERR_STATE,
                          "emdee_dd: domain %d re-sorted %d atoms, its messages say %d + %d", d.geo.rank, d.host_w[DDW_NLIVE], d.host_w[DDW_NNEW], d.host_w[DDW_NGHOST]);
            sy.commit_edit(d.host_w[DDW_NLIVE]);
            d.n_owned = d.host_w[DDW_NNEW];
            stat_migrated += d.host_w[DDW_NLEAVE];
            d.plan = DdPlan{};
            d.plan.npeers = np;
            for (int p = 0; p < np; p++) {
                d.plan.send_start[p + 1] = d.plan.send_start[p] + d.host_w[DDW_GSEND + p];
                d.plan.recv_start[p + 1] = d.plan.recv_start[p] + d.host_w[DDW_GRECV + p];
            }
            for (int p = 0; p <= np; p++) d.plan.ghost_id[p] = e.ghost_base + d.gr_caps.start[p];   // (the capacities THIS rebuild ran with)
            d.n_send = d.plan.send_start[np];
            d.n_ghost = d.plan.recv_start[np];
            EMDEE_REQUIRE(d.n_send == d.host_w[DDW_NSEND] && d.n_ghost == d.host_w[DDW_NGHOST], EMDEE_ERR_STATE, "emdee_dd: ghost counts inconsistent");
            d.md->n_ghost = d.n_ghost;
            d.md->since_build = 0;
            d.md->current_mask = 0;
            if (!built[l]) {
                // an engine on the direct kernels (they count atoms on the host): its re-sorted state is complete but has no list --
                // a load from its own state, which nobody else needs to know about; the send list follows the new numbering
                export_caller_arrays(d, d.ids.ptr, d.n_send);
                load_engine(d, with_forces);
                continue;
            }
            if (with_forces && sy.n_total > 0) { d.md->forces(EMDEE_FORCES, 0); }
            set_caps(d);
            finish_rebuild(d);
        }
        stat_rebuilds++;
        stat_fast++;
        return true;
    }
    int64_t stat_regrown = 0, stat_rebuild_readbacks = 0;

    static void grow_keep(DevBuf<real> &b, size_t keep, size_t want, hipStream_t s) { grow_keep_t(b, keep, want, s); }
    static void grow_keep(DevBuf<emdee_lj_atom> &b, size_t keep, size_t want, hipStream_t s) { grow_keep_t(b, keep, want, s); }
    static void grow_keep(DevBuf<long long> &b, size_t keep, size_t want, hipStream_t s) { grow_keep_t(b, keep, want, s); }
    template <typename T>
    static void grow_keep_t(DevBuf<T> &b, size_t keep, size_t want, hipStream_t s) {
        if (want <= b.cap) return;
        DevBuf<T> nb;
        nb.ensure(want);
        if (keep) EMDEE_HIP_CHECK(hipMemcpyAsync(nb.ptr, b.ptr, keep * sizeof(T), hipMemcpyDeviceToDevice, s));
        EMDEE_HIP_CHECK(hipStreamSynchronize(s));
        b.swap(nb);
    }

    // ---------------------------------------------------------------- one halo exchange around a compute call
    // pack (request word = *V) -> exchange || compute(1) -> unpack (G |= requests) -> compute(2)
    // The overlapped form of one process = one domain, in LOCK STEP on two streams (round 3).  Round 2 sent a step through
    // three streams -- pack on the compute stream, the messages on a communication stream, unpack + boundary bricks on a
    // third -- i.e. three event hops on the critical path of every step (pack -> messages -> boundary half -> next pack),
    // each ~10 us of empty queue: the overlapped form lost to the in-order one on every one-GPU rehearsal.  Here the whole
    // halo half of a step -- pack, ncclSend/ncclRecv, unpack, boundary bricks -- is queued IN ORDER on the halo stream, the
    // interior bricks on the compute stream, and the two meet once per step: the halo half waits for what the compute
    // stream had produced when the step began (interior launch of the previous step: positions, its rebuild request word;
    // the thermostat's noise), the interior launch for the boundary half of the previous step.  Two records and two waits,
    // one hop on the critical path.
    bool lockstep_ok() const {
        return overlap && two_streams && lockstep && !in_rebuild && dom.size() == 1 && (use_rccl || mirror || dom[0]->geo.npeers == 0);
    }
    // the compute stream catches up with the halo stream (before a read-back, a rebuild, anything that is not a step)
    void join_halo() {
        if (!halo_live) return;
        EMDEE_HIP_CHECK(hipStreamWaitEvent(dom[0]->ctx->stream, dom[0]->ev_bnd, 0));
        halo_live = false;
    }
    template <class F>
    void with_halo_lockstep(int vj, int gj, F &&compute) {
        const size_t w = sizeof(real);
        Domain<real> &d = *dom[0];
        hipStream_t main_stream = d.ctx->stream;
        EMDEE_HIP_CHECK(hipEventRecord(d.ev_packed, main_stream));           // everything up to the previous interior launch
        if (halo_live) EMDEE_HIP_CHECK(hipStreamWaitEvent(main_stream, d.ev_bnd, 0));   // boundary half of the previous step
        compute(d, 1);
        struct Restore {
            emdee_ctx *ctx;
            hipStream_t keep;
            bool &flag;
            ~Restore() { ctx->stream = keep; flag = false; }
        } restore{d.ctx, main_stream, force_inline};
        d.ctx->stream = d.side;
        force_inline = true;
        EMDEE_HIP_CHECK(hipStreamWaitEvent(d.side, d.ev_packed, 0));
        const int np = d.geo.npeers;
        const bool timed = d.sys().profiling;
        // (the pair is closed whatever happens in between: an exchange that throws must not leave an end event that was never
        // recorded -- every later hipEventElapsedTime of the timer would fail)
        struct HaloPair {
            KernelTimer *t;
            size_t k;
            hipStream_t s;
            ~HaloPair() { if (t) (void)hipEventRecord(t->pairs[k].second, s); }
            void end() { if (t) { KernelTimer *q = t; t = nullptr; q->end(k, s); } }
        } halo_pair{nullptr, 0, d.side};
        const size_t tk = timed ? d.sys().timers[T_HALO].begin(d.side) : 0;
        if (timed) { halo_pair.t = &d.sys().timers[T_HALO]; halo_pair.k = tk; d.halo_batch_timed = true; }
        hipLaunchKernelGGL((k_dd_pack_step<real>), dim3(blocks_for(std::max(d.n_send, std::max(np, 1)), 256)), dim3(256), 0, d.stream(),
                           d.n_send, d.plan, d.ids.ptr, d.codes.ptr, d.geo.template device<real>(), d.sys().inv_perm.ptr, d.sys().rec.ptr,
                           d.V(vj), d.sendbuf.ptr);
        for (int p = 0; p < np; p++) {
            d.xf.soff[p] = dd_msg_begin(d.plan.send_start, p, w); d.xf.sbytes[p] = dd_msg_bytes(d.plan.send_start, p, w);
            d.xf.roff[p] = dd_msg_begin(d.plan.recv_start, p, w); d.xf.rbytes[p] = dd_msg_bytes(d.plan.recv_start, p, w);
        }
        d.xf.send = d.sendbuf.ptr;
        d.xf.recv = d.recvbuf.ptr;
        exchange();                                                        // in order on the halo stream
        hipLaunchKernelGGL((k_dd_unpack_step<real>), dim3(blocks_for(std::max(d.n_ghost, std::max(np, 1)), 256)), dim3(256), 0, d.stream(),
                           d.n_ghost, d.n_owned, d.plan, d.sys().inv_perm.ptr, d.recvbuf.ptr, d.sys().rec.ptr, d.V(vj), d.G(gj));
        halo_pair.end();
        d.md->current_mask = 0;
        compute(d, 2);
        EMDEE_HIP_CHECK(hipEventRecord(d.ev_bnd, d.side));
        halo_live = true;
    }

    template <class F>
    void with_halo(int vj, int gj, F &&compute) {
        if (lockstep_ok()) {
            with_halo_lockstep(vj, gj, compute);
            return;
        }
        join_halo();
        const size_t w = sizeof(real);
        struct ClosePairs {                                        // (as in the lock-step form: no half-recorded T_HALO pair survives an exception)
            DdImpl *dd;
            ~ClosePairs() {
                for (auto &pd : dd->dom)
                    if (pd->halo_timed) { pd->halo_timed = false; (void)hipEventRecord(pd->sys().timers[T_HALO].pairs[pd->halo_tk].second, pd->stream()); }
            }
        } close_pairs{this};
        for (auto &pd : dom) {
            Domain<real> &d = *pd;
            const int np = d.geo.npeers;
            const int nthreads = std::max(d.n_send, std::max(np, 1));
            d.halo_timed = d.sys().profiling;
            if (d.halo_timed) { d.halo_tk = d.sys().timers[T_HALO].begin(d.stream()); d.halo_batch_timed = true; }
            hipLaunchKernelGGL((k_dd_pack_step<real>), dim3(blocks_for(nthreads, 256)), dim3(256), 0, d.stream(), d.n_send, d.plan,
                               d.ids.ptr, d.codes.ptr, d.geo.template device<real>(), d.sys().inv_perm.ptr, d.sys().rec.ptr, d.V(vj),
                               d.sendbuf.ptr);
            for (int p = 0; p < np; p++) {
                d.xf.soff[p] = dd_msg_begin(d.plan.send_start, p, w); d.xf.sbytes[p] = dd_msg_bytes(d.plan.send_start, p, w);
                d.xf.roff[p] = dd_msg_begin(d.plan.recv_start, p, w); d.xf.rbytes[p] = dd_msg_bytes(d.plan.recv_start, p, w);
            }
            d.xf.send = d.sendbuf.ptr;
            d.xf.recv = d.recvbuf.ptr;
            record_packed(d);
        }
        exchange();
        if (overlap)
            for (auto &pd : dom) compute(*pd, 1);
        wait_exchange();
        for (auto &pd : dom) {
            Domain<real> &d = *pd;
            // The boundary half of the step -- unpack, then the launch over the boundary bricks -- goes to a stream of its
            // own: it needs the halo, not the interior launch, so it starts as soon as the messages are in and fills the
            // CUs the interior launch's last workgroups leave idle (one rank's box of the 8-rank 10^7-atom run, timed
            // alone: two launches back to back 0.212 ms, one launch over all bricks 0.177 ms; profiles/dd_rank_proxy.py).
            // Everything the engine queues meanwhile follows ctx->stream, which is pointed at that stream for the duration.
            const bool aside = overlap && two_streams;
            hipStream_t main_stream = d.ctx->stream;
            {
                // (in RCCL mode d.ctx is the CALLER's context: whatever is thrown below, its stream is put back)
                struct StreamSwap {
                    emdee_ctx *ctx;
                    hipStream_t keep;
                    ~StreamSwap() { ctx->stream = keep; }
                } swap_back{d.ctx, main_stream};
                if (aside) {
                    EMDEE_HIP_CHECK(hipStreamWaitEvent(d.side, d.ev_packed, 0));   // all earlier work of this domain
                    EMDEE_HIP_CHECK(hipStreamWaitEvent(d.side, d.ev_done, 0));     // the halo
                    d.ctx->stream = d.side;
                }
                const int nthreads = std::max(d.n_ghost, std::max(d.geo.npeers, 1));
                hipLaunchKernelGGL((k_dd_unpack_step<real>), dim3(blocks_for(nthreads, 256)), dim3(256), 0, d.stream(), d.n_ghost, d.n_owned,
                                   d.plan, d.sys().inv_perm.ptr, d.recvbuf.ptr, d.sys().rec.ptr, d.V(vj), d.G(gj));
                if (d.halo_timed) { d.halo_timed = false; d.sys().timers[T_HALO].end(d.halo_tk, d.stream()); }
                d.md->current_mask = 0;
                compute(d, overlap ? 2 : 0);
                if (aside) EMDEE_HIP_CHECK(hipEventRecord(d.ev_bnd, d.side));
            }
            if (aside) EMDEE_HIP_CHECK(hipStreamWaitEvent(main_stream, d.ev_bnd, 0));   // the next step (and any read-back) sees both halves
        }
    }

    // ---------------------------------------------------------------- stepping
    void set_langevin(double gamma, double temperature, uint64_t seed, uint64_t first_step) override {
        lgv_on = gamma > 0.0;
        lgv_gamma = gamma; lgv_T = temperature; lgv_seed = seed; lgv_first = first_step;
        for (auto &d : dom) {
            d->md->set_langevin(gamma, temperature, seed, first_step);   // (the noise is keyed by the global ids that travel with the atoms)
        }
    }

    bool read_global_words(int first, int count, int *out) {
        // identical on every domain by construction: read the first local one (debug builds could compare)
        Domain<real> &d = *dom[0];
        join_halo();
        if (dom.size() == 1) {
            read_back_words(d.ctx, d.stream(), d.G(first), count, out);
        } else {
            EMDEE_HIP_CHECK(hipMemcpyAsync(d.ctx->host_flags, d.G(first), count * sizeof(int), hipMemcpyDeviceToHost, d.stream()));
            for (auto &pd : dom) EMDEE_HIP_CHECK(hipStreamSynchronize(pd->stream()));
            for (int k = 0; k < count; k++) out[k] = d.ctx->host_flags[k];
        }
        bool any = false;
        for (int k = 0; k < count; k++) any = any || out[k] != 0;
        return any;
    }

    // plain force pass (all outputs in `bitmask`) at the current positions with fresh ghosts; true if the list was stale
    // for them (then a rebuild has been done and the forces recomputed)
    void forces_with_halo(int bitmask, int carry, bool check_displacement = true) {
        for (auto &pd : dom) hipLaunchKernelGGL(k_dd_batch_begin, dim3(1), dim3(64), 0, pd->stream(), pd->words.ptr, DD_WORDS, carry);
        with_halo(0, 0, [&](Domain<real> &d, int phase) { d.md->forces(bitmask, phase); });
        join_halo();
        int g = 0;
        if (check_displacement && read_global_words(0, 1, &g)) {
            redistribute(true);
            if (bitmask != EMDEE_FORCES)
                for (auto &pd : dom) pd->md->forces(bitmask, 0);
        }
    }

    // The inner step that follows a rebuild in the middle of a run, at the positions the rebuild sorted: ONE fused launch
    // over all bricks (force + kick + drift; the ghosts are fresh, nobody is waited for), which raises V[0] for the positions
    // it produces -- where round 2 ran a force pass and a separate kick + drift (19 us more per rebuild of a 1.26 M-atom
    // rank, and two fills).  Domains on the direct kernels keep the split form.
    void step_after_rebuild(double dt) {
        for (auto &pd : dom) {
            Domain<real> &d = *pd;
            // (a rebuild in the engine's order has cleared the words with its first launch: nothing between its read-back and the step)
            if (!d.words_clear) EMDEE_HIP_CHECK(hipMemsetAsync(d.words.ptr, 0, DD_WORDS * sizeof(int), d.stream()));
            d.words_clear = false;
            if (d.sys().n_total > 0) {
                const bool tiled = d.sys().brick_active && !(d.md->current_mask & EMDEE_FORCES) &&
                                   d.sys().fused_step(dt, dt, 0, nullptr, d.V(0), false, false);
                if (!tiled) {
                    if (!(d.md->current_mask & EMDEE_FORCES)) d.md->forces(EMDEE_FORCES, 0);
                    d.sys().kick_drift(dt, dt, d.V(0));
                }
            }
            d.md->current_mask = 0;
        }
    }

    void step(int nsteps, double dt, int rebuild_every) override {
        use_device(user_ctx);
        EMDEE_REQUIRE(loaded, EMDEE_ERR_STATE, "emdee_dd_step: call emdee_dd_load first");
        EMDEE_REQUIRE(nsteps >= 0 && dt >= 0 && rebuild_every >= 0, EMDEE_ERR_INVALID, "emdee_dd_step: negative argument");
        if (nsteps == 0) return;
        // Which kernels a domain steps with is ITS business and may change at any rebuild (brick_active: the densest tile of
        // this domain fits LDS or not; an empty domain launches nothing): the batches, their exchanges and the guard words
        // are the same for everybody, so the send/recv sequences of the ranks cannot drift apart.
        for (auto &pd : dom) {
            Domain<real> &d = *pd;
            if (!(d.md->current_mask & EMDEE_FORCES)) EMDEE_REQUIRE(false, EMDEE_ERR_STATE, "emdee_dd_step: forces are not current");
            EMDEE_HIP_CHECK(hipMemsetAsync(d.words.ptr, 0, DD_WORDS * sizeof(int), d.stream()));
            d.sys().kick_drift(0.5 * dt, dt, d.V(0));       // x_1 = x_0 + dt (v_0 + dt/2 f_0)
            d.md->current_mask = 0;
        }
        int s = 1, carry = 0;                               // carry: index of the word that flags the current positions
        while (s < nsteps) {
            // ---- a batch of inner steps: force + full kick + drift in one kernel pass each
            int B = std::min(max_batch, nsteps - s);
            if (rebuild_every > 0) B = std::min(B, std::max(1, rebuild_every - dom[0]->since_build - 1));
            // displacement trigger: queue what the previous interval between rebuilds says is safe, then one step at a
            // time until the request comes -- a cancelled step costs a halo exchange that nobody uses
            else B = std::max(1, std::min(B, last_interval > 0 ? last_interval - dom[0]->since_build - 1 : 2));
            if (rebuild_every > 0 && dom[0]->since_build + 1 >= rebuild_every) {
                // fixed cadence: rebuild at the current positions, then the un-fused equivalent of one inner step
                redistribute(true, false);
                step_after_rebuild(dt);
                for (auto &pd : dom) pd->since_build = 0;
                carry = 0;
                s++;
                continue;
            }
            for (auto &pd : dom) {
                pd->halo_batch_timed = false;
                hipLaunchKernelGGL(k_dd_batch_begin, dim3(1), dim3(64), 0, pd->stream(), pd->words.ptr, DD_WORDS, carry);
            }
            // Waiting for the request (the steps the last interval vouched for are done, single steps follow): these steps run IN
            // ORDER -- halo first, then one launch over all bricks under everybody's word.  Overlapped, the interior launch starts
            // under the domain's OWN word only, and when the request comes from a neighbour -- for seven of eight ranks it does --
            // that launch is void: ~0.1 ms per rebuild of a 1.26 M-atom rank thrown away, against a halo left uncovered for the one
            // or two steps concerned.  Same states either way (EMDEE_DD_HOLD_INTERIOR=0: overlapped throughout, as rounds 3-4).
            const bool waiting = hold_interior && overlap && rebuild_every == 0 && last_interval > 0 &&
                                 last_interval - dom[0]->since_build - 1 <= 0 && (world > 1 || dom.size() > 1);
            struct OverlapBack { bool &o; bool keep; ~OverlapBack() { o = keep; } } overlap_back{overlap, overlap};
            if (waiting) { join_halo(); overlap = false; }
            for (int j = 0; j < B; j++) {
                if (lgv_on) join_halo();   // (the boundary half of the previous step still reads the previous noise)
                for (auto &pd : dom)
                    if (pd->sys().brick_active) pd->sys().prepare_noise(dt);   // (thermostat only) before the pack: both halves read it
                with_halo(j, j, [&](Domain<real> &d, int phase) {
                    if (d.sys().n_total == 0) return;                     // nothing to move; its words stay clear
                    if (d.sys().brick_active) {
                        // interior bricks look at my own request only (their neighbours are all mine); boundary bricks
                        // at the OR of everybody's
                        const int *guard = (phase == 1) ? d.V(j) : d.G(j);
                        const bool launched = d.sys().fused_step(dt, dt, phase, rebuild_every > 0 ? nullptr : guard, d.V(j + 1), false, true);
                        EMDEE_REQUIRE(launched, EMDEE_ERR_STATE, "emdee_dd_step: domain %d could not launch its step kernel", d.geo.rank);
                    } else if (phase != 1) {
                        // tiles too large for LDS (dense slab, long cutoff): the direct kernels, whole domain behind the halo,
                        // under the same words
                        d.sys().guarded_split_step(dt, dt, rebuild_every > 0 ? nullptr : d.G(j), d.V(j + 1));
                    }
                });
            }
            join_halo();
            int g[DD_MAX_BATCH];
            int ran = B;
            stat_batches++;
            if (rebuild_every == 0 && read_global_words(0, B, g)) {
                for (int j = 0; j < B; j++)
                    if (g[j]) { ran = j; break; }
                stat_cancelled += B - ran;
            }
            for (auto &pd : dom) {
                Domain<real> &d = *pd;
                const bool tiled = d.sys().brick_active && d.sys().n_total > 0;
                if (tiled && ((B - ran) & 1)) d.sys().swap_step_buffers();   // the cancelled launches did not advance the ping-pong
                if (d.sys().lgv_on && d.sys().n_total > 0) d.sys().lgv_step -= (unsigned long long)(B - ran);
                if (tiled && d.sys().profiling) {
                    d.sys().timers[T_STEP].dropped += B - ran;
                    if (overlap) d.sys().timers[T_STEP_BOUNDARY].dropped += B - ran;
                    if (d.halo_batch_timed) d.sys().timers[T_HALO].dropped += B - ran;   // (only steps whose halo was timed)
                }
                if (!tiled && d.sys().profiling && d.sys().n_total > 0) {
                    d.sys().timers[T_FORCE].dropped += B - ran;
                    d.sys().timers[T_KICK_DRIFT].dropped += B - ran;
                }
                d.since_build += ran;
                d.md->current_mask = 0;
            }
            s += ran;
            carry = ran;                                     // V[ran]: raised by step ran - 1 for the positions it produced
            if (ran < B) {
                // the positions of step s were flagged: rebuild there (evaluates the forces), then the un-fused
                // equivalent of that inner step
                last_interval = dom[0]->since_build;          // steps the list just retired has served
                redistribute(true, s >= nsteps);
                if (s < nsteps) {
                    step_after_rebuild(dt);
                    carry = 0;
                    s++;
                }
            }
        }
        join_halo();
        // ---- last step: plain force pass and the closing half kick
        if (dom[0]->md->current_mask & EMDEE_FORCES) {
            // (a rebuild at the last positions has just evaluated them)
        } else if (rebuild_every > 0 && dom[0]->since_build + 1 >= rebuild_every) {
            redistribute(true);
        } else {
            forces_with_halo(EMDEE_FORCES, carry, rebuild_every == 0);   // (a fixed cadence does not look at displacements)
        }
        for (auto &pd : dom) {
            pd->sys().kick(0.5 * dt);
            pd->md->current_mask = EMDEE_FORCES;
            EMDEE_HIP_CHECK(hipMemsetAsync(pd->words.ptr, 0, DD_WORDS * sizeof(int), pd->stream()));
        }
        EMDEE_HIP_CHECK(hipGetLastError());
    }

    // ---------------------------------------------------------------- state out
    void energies(double out[3]) override {
        use_device(user_ctx);
        EMDEE_REQUIRE(loaded, EMDEE_ERR_STATE, "emdee_dd_energies: call emdee_dd_load first");
        join_halo();
        std::vector<std::vector<double>> vals;
        for (auto &pd : dom) {
            // ghosts are current whenever the forces are (every force pass follows a halo unpack or a rebuild)
            double e[3];
            pd->md->energies(e);
            vals.push_back({e[0], e[1], e[2]});
        }
        allreduce_sum(vals, 3, out);
    }
    int64_t n_atoms_global() override { return n_global; }
    int n_owned(int l) override { return local(l).n_owned; }
    int n_ghost(int l) override { return local(l).n_ghost; }
    IMd *engine(int l) override { return local(l).md.get(); }
    void get_state(int l, int64_t *gids, void *pos, void *vel, void *frc) override {
        use_device(user_ctx);
        Domain<real> &d = local(l);
        EMDEE_REQUIRE(loaded, EMDEE_ERR_STATE, "emdee_dd_get_state: call emdee_dd_load first");
        join_halo();
        const size_t n = (size_t)d.n_owned, nt = n + (size_t)d.n_ghost;
        // caller-order copies of the whole domain (positions and ids include the ghosts) into scratch, owned part out
        d.f.ensure(3 * n + 3); d.x2.ensure(3 * nt + 3); d.v2.ensure(3 * n + 3); d.gid2.ensure(nt + 1);
        d.sys().unsort(d.x2.ptr, d.v2.ptr, d.f.ptr, nullptr, nullptr, nullptr, d.gid2.ptr);
        FenceOut fence(user_ctx, d.stream());
        hipStream_t s = d.stream();
        if (n > 0) {
            if (pos) EMDEE_HIP_CHECK(hipMemcpyAsync(pos, d.x2.ptr, 3 * n * sizeof(real), hipMemcpyDeviceToDevice, s));
            if (vel) EMDEE_HIP_CHECK(hipMemcpyAsync(vel, d.v2.ptr, 3 * n * sizeof(real), hipMemcpyDeviceToDevice, s));
            if (frc) EMDEE_HIP_CHECK(hipMemcpyAsync(frc, d.f.ptr, 3 * n * sizeof(real), hipMemcpyDeviceToDevice, s));
            if (gids) EMDEE_HIP_CHECK(hipMemcpyAsync(gids, d.gid2.ptr, n * sizeof(long long), hipMemcpyDeviceToDevice, s));
        }
    }
    void stats(int64_t out[4]) override {
        out[0] = stat_rebuilds; out[1] = stat_batches; out[2] = stat_cancelled; out[3] = stat_migrated;
    }
    // host wall-clock of the rebuilds (ownership path, exchanges, read-backs, the engines' sort + list: everything between
    // the last step before and the first step after) and of the blocking read-backs, cumulative; ghost share of domain 0
    double stat_rebuild_wall_ms = 0.0;
    int64_t stat_rebuild_calls = 0;
    void phase_times(double out[8]) override {
        for (int k = 0; k < 8; k++) out[k] = 0.0;
        out[0] = stat_rebuild_wall_ms; out[1] = (double)stat_rebuild_calls;
        double rb = 0.0, nrb = 0.0;
        std::vector<const emdee_ctx *> seen;
        for (auto &pd : dom) {
            if (std::find(seen.begin(), seen.end(), pd->ctx) != seen.end()) continue;
            seen.push_back(pd->ctx);
            rb += pd->ctx->readback_ms; nrb += (double)pd->ctx->readbacks;
        }
        out[2] = rb; out[3] = nrb;
        if (!dom.empty() && dom[0]->n_owned + dom[0]->n_ghost > 0) out[4] = (double)dom[0]->n_ghost / (double)(dom[0]->n_owned + dom[0]->n_ghost);
        out[5] = (double)stat_rebuild_readbacks; out[6] = (double)stat_fast; out[7] = (double)stat_regrown;
    }
    void rebuild_stats(int64_t out[4]) override {
        out[0] = stat_fast + stat_fallback; out[1] = stat_fallback;
        out[2] = dom.empty() || dom[0]->geo.npeers == 0 ? 0 : dom[0]->mig_caps.start[1];
        out[3] = dom.empty() ? 0 : dom[0]->gs_caps.start[dom[0]->geo.npeers];
    }
    void set_overlap(bool on) override {
        join_halo();
        for (auto &pd : dom) EMDEE_HIP_CHECK(hipStreamSynchronize(pd->stream()));
        overlap = on;
    }
};

template <typename real>
IDd *Factory<real>::dd(emdee_ctx *ctx, const double len[3], const int32_t grid[3], int rank_first, int n_local,
                       const void *unique_id, const emdee_lj_model &model, double skin) {
    const int g[3] = {grid[0], grid[1], grid[2]};
    return new DdImpl<real>(ctx, len, g, rank_first, n_local, unique_id, model, skin);
}

}  // namespace emdee
