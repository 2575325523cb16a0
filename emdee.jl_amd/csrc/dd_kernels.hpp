// dd_kernels.hpp -- device side of the spatial domain decomposition (absent from the
// reference, which is single-GPU): ownership and migration, ghost selection, and the per-step halo
// messages.  Everything here is integer/byte work over the atoms of ONE domain; the transports that
// move the packed messages between domains are in dd.hpp.
//
// Geometry (same conventions as emdee.jl_amd/domain.py, whose CPU/gloo tests pin them): the global box
// [0, L_d) is cut into g_x x g_y x g_z bricks, rank = cx + g_x (cy + g_y cz).  Along a cut dimension a
// rank's local box is its brick plus a halo of width h = cutoff + skin filled with ghost atoms, images
// shifted by -+L_d where the neighbour wraps around the global box.  Full (owner-computes) lists: only
// positions travel, no force is sent back.
#pragma once

#include "kernels.hpp"

namespace emdee {

constexpr int DD_MAX_DIRS = 26;
constexpr int DD_MAX_PEERS = 26;
constexpr int DD_MAX_WORLD = 64;
constexpr int DD_HDR = 16;            // bytes in front of every per-step halo message (word 0 = rebuild request)

// What one domain knows about the decomposition, passed to kernels by value.
template <typename real>
struct DdDev {
    real L[3], width[3], lo[3], hi[3], halo;
    int grid[3], cut[3];
    int rank, world;
    int ndirs;
    int dir[DD_MAX_DIRS][3];          // neighbour directions (-1, 0, +1 per dimension), only cut dimensions non-zero
    real shift[DD_MAX_DIRS][3];       // periodic image shift a ghost sent in that direction carries
    int dir_bin[DD_MAX_DIRS];         // position of the direction in the send list: sorted by (destination rank, direction)
    int bin_dir[DD_MAX_DIRS];         // ... and back
    int rank_bin[DD_MAX_WORLD];       // migration: destination rank -> bin (0 = stays, 1 + peer index, -1 = not a neighbour)
    real mig_off[DD_MAX_PEERS + 1][3]; // subtracted from the position of an atom that leaves in bin b: zero, except in the replica
                                      // rehearsal of one rank (dd.hpp, DdImpl::mirror), where the leaver comes back in as its own image
};

// rows that travel at a rebuild
template <typename real>
struct MigRow {                       // an atom changing owner
    real x[3], v[3];
    float hs, te;
    long long gid;
};
template <typename real>
struct GhostRow {                     // a ghost as first sent: shifted position + LJAtom + global id (the engines order a cell's
    real x[3];                        // atoms by it, ghosts included: NbSystem::tag)
    float hs, te;
    long long gid;
};

// prefix of atoms per peer for the per-step messages (message p = DD_HDR bytes + 3 reals per atom)
struct DdPlan {
    int npeers;
    int send_start[DD_MAX_PEERS + 1];
    int recv_start[DD_MAX_PEERS + 1];
    int ghost_id[DD_MAX_PEERS + 1];   // engine id of the first ghost received from peer p (the ids of a peer's ghosts are consecutive)
};
// message p begins at byte p DD_HDR + 3 w start[p] and holds DD_HDR + 3 w (start[p+1] - start[p]) bytes
static inline size_t dd_msg_begin(const int *start, int p, size_t w) { return (size_t)p * DD_HDR + (size_t)start[p] * 3 * w; }
static inline size_t dd_msg_bytes(const int *start, int p, size_t w) { return DD_HDR + (size_t)(start[p + 1] - start[p]) * 3 * w; }

// ------------------------------------------------------------------------------------ stable multi-bin partition
// Items carry a bit mask of bins (<= 32 bins; an item may sit in several: a corner atom is a ghost of up to
// seven neighbours).  Output: for every bin, the ids of its items in ascending order -- deterministic, no
// atomics on the output.  count -> exclusive scan of counts[bin][block] -> scatter.  The kernels that produce the
// masks (ownership, ghost selection) count their block themselves: blocks of PART_BLOCK items.
constexpr int PART_BLOCK = 1024;     // (1024: the count arrays of a rank-sized domain -- bins x blocks -- fit one launch of k_scan_block)

struct DdBins {                      // the bins of a partition that belong to peer p: lo[p] .. lo[p+1]-1
    int npeers;
    int lo[DD_MAX_PEERS + 2];
};
// the scanned counts of a partition, as its consumers read them: first output position of bin b = offs[b * nblocks]
struct PartView {
    const int *offs;
    int nblocks;
    __device__ __forceinline__ int start(int bin) const { return offs[(size_t)bin * nblocks]; }
};

// every thread of the block calls this with the mask of its item (0 past the end)
__device__ __forceinline__ void part_count_block(unsigned m, int nbins, int nblocks, int *__restrict__ counts) {
    __shared__ int c[32];
    if (threadIdx.x < 32) c[threadIdx.x] = 0;
    __syncthreads();
    const int lane = threadIdx.x & (WAVE - 1);
    for (int b = 0; b < nbins; b++) {
        const unsigned long long bal = __ballot((m >> b) & 1u);
        if (lane == 0 && bal) atomicAdd(&c[b], __popcll(bal));
    }
    __syncthreads();
    if ((int)threadIdx.x < nbins) counts[(size_t)threadIdx.x * nblocks + blockIdx.x] = c[threadIdx.x];
}

// ------------------------------------------------------------------------------------ ownership
// coordinate p wrapped into [0, L): ONE function for every ownership pass, so that the rebuild paths agree to the bit
template <typename real>
__device__ __forceinline__ real dd_wrap(real p, real L) { return p - L * floor(p / L); }

// Wrap every owned atom into the global box and name the rank whose brick contains it.
// mask[i] = 1 << bin: bin 0 = stays here, 1 + p = leaves for peer p.  An atom that would have to
// jump over a brick (cannot happen while the halo exceeds the displacement between rebuilds) raises *err.
template <typename real>
__global__ __launch_bounds__(PART_BLOCK) void k_dd_classify(int n, real *__restrict__ x, DdDev<real> g, unsigned *__restrict__ mask,
                                                            int *__restrict__ err, int nbins, int nblocks, int *__restrict__ counts) {
    const int i = blockIdx.x * PART_BLOCK + threadIdx.x;
    unsigned m = 0;
    if (i < n) {
        int c[3];
#pragma unroll
        for (int d = 0; d < 3; d++) {
            const real p = dd_wrap(x[3 * (size_t)i + d], g.L[d]);
            x[3 * (size_t)i + d] = p;
            c[d] = min(max((int)floor(p / g.width[d]), 0), g.grid[d] - 1);
        }
        const int dest = c[0] + g.grid[0] * (c[1] + g.grid[1] * c[2]);
        int bin = g.rank_bin[dest];
        if (bin < 0) { *err = 1; bin = 0; }
        m = 1u << bin;
        mask[i] = m;
    }
    part_count_block(m, nbins, nblocks, counts);
}

// Which neighbours need this owned atom as a ghost: bit dir_bin[k] for every direction k whose halo holds it.
// (n_dev, optional: the number of items as a device word -- a count-free rebuild launches over an upper bound)
template <typename real>
__global__ __launch_bounds__(PART_BLOCK) void k_dd_ghost_mask(int n, const real *__restrict__ x, DdDev<real> g, unsigned *__restrict__ mask,
                                                              int nbins, int nblocks, int *__restrict__ counts,
                                                              const int *__restrict__ n_dev = nullptr) {
    const int i = blockIdx.x * PART_BLOCK + threadIdx.x;
    unsigned m = 0;
    if (n_dev) n = min(n, *n_dev);
    if (i < n) {
        bool near_lo[3], near_hi[3];
#pragma unroll
        for (int d = 0; d < 3; d++) {
            const real p = x[3 * (size_t)i + d];
            near_lo[d] = g.cut[d] && p < g.lo[d] + g.halo;
            near_hi[d] = g.cut[d] && p >= g.hi[d] - g.halo;
        }
        for (int k = 0; k < g.ndirs; k++) {
            bool in = true;
#pragma unroll
            for (int d = 0; d < 3; d++) {
                const int s = g.dir[k][d];
                in = in && (s == 0 || (s > 0 ? near_hi[d] : near_lo[d]));
            }
            if (in) m |= 1u << g.dir_bin[k];
        }
        mask[i] = m;
    }
    part_count_block(m, nbins, nblocks, counts);
}

static __global__ __launch_bounds__(PART_BLOCK) void k_part_scatter(int n, const unsigned *__restrict__ mask, int nbins,
                                                                    int nblocks, const int *__restrict__ offs,
                                                                    int *__restrict__ out_id, int *__restrict__ out_bin,
                                                                    const int *__restrict__ n_dev = nullptr, int out_cap = 0x7fffffff) {
    __shared__ int wc[PART_BLOCK / WAVE][32];
    const int i = blockIdx.x * PART_BLOCK + threadIdx.x;
    if (n_dev) n = min(n, *n_dev);
    const unsigned m = i < n ? mask[i] : 0u;
    const int lane = threadIdx.x & (WAVE - 1), wv = threadIdx.x / WAVE;
    for (int b = 0; b < nbins; b++) {
        const unsigned long long bal = __ballot((m >> b) & 1u);
        if (lane == 0) wc[wv][b] = __popcll(bal);
    }
    __syncthreads();
    for (int b = 0; b < nbins; b++) {
        const unsigned long long bal = __ballot((m >> b) & 1u);
        if ((m >> b) & 1u) {
            int pos = offs[(size_t)b * nblocks + blockIdx.x];
            for (int w = 0; w < wv; w++) pos += wc[w][b];
            pos += prefix_popc(bal);
            if (pos < out_cap) {                               // (a count-free rebuild whose lists outgrew their capacity is redone)
                out_id[pos] = i;
                if (out_bin) out_bin[pos] = b;
            }
        }
    }
}

// bin_start[0..nbins] from the scanned counts; peer_count[p] = entries bound for peer p (bins lo[p] .. lo[p+1]-1)
static __global__ void k_part_starts(int nbins, int nblocks, const int *__restrict__ offs, int *__restrict__ bin_start,
                                     DdBins pb, int *__restrict__ peer_count) {
    const int t = threadIdx.x;
    if (t <= nbins) bin_start[t] = offs[(size_t)t * nblocks];
    if (t < pb.npeers) peer_count[t] = offs[(size_t)pb.lo[t + 1] * nblocks] - offs[(size_t)pb.lo[t] * nblocks];
}

// the counted rebuild: my error word rides on the counts I send (bit 30 of every per-peer count), so that all ranks see it
// in the exchange of counts and fail together
constexpr int DD_COUNT_ERR = 1 << 30;
static __global__ void k_dd_flag_counts(int npeers, const int *__restrict__ err, int *__restrict__ peer_count) {
    const int t = threadIdx.x;
    if (t < npeers && *err != 0) peer_count[t] |= DD_COUNT_ERR;
}
// engine ids -> positions in the dense caller-order arrays (NbSystem::ids_map)
static __global__ void k_dd_map_ids(int n, const int *__restrict__ map, int *__restrict__ ids) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < n) ids[k] = map[ids[k]];
}

// ------------------------------------------------------------------------------------ migration rows
template <typename real>
__global__ void k_dd_pack_migrants(int first, int total, const int *__restrict__ ids, const real *__restrict__ x,
                                   const real *__restrict__ v, const emdee_lj_atom *__restrict__ atoms,
                                   const long long *__restrict__ gid, MigRow<real> *__restrict__ rows, DdDev<real> g, DdBins starts) {
    int k = first + blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= total) return;
    const int i = ids[k];
    int bin = 1;                       // starts.lo[b] = first list position of bin b (the leavers of peer b - 1)
    while (bin < starts.npeers && k >= starts.lo[bin + 1]) bin++;
    MigRow<real> r;
#pragma unroll
    for (int d = 0; d < 3; d++) { r.x[d] = x[3 * (size_t)i + d] - g.mig_off[bin][d]; r.v[d] = v[3 * (size_t)i + d]; }
    r.hs = atoms[i].half_sigma; r.te = atoms[i].twice_sqrt_eps;
    r.gid = gid[i];
    rows[k - first] = r;
}

// new owned arrays: the atoms that stay (ids[0 .. n_stay), ascending) followed by the arrivals (ordered by source rank)
template <typename real>
__global__ void k_dd_assemble_owned(int n_stay, int n_arrive, const int *__restrict__ ids, const real *__restrict__ x,
                                    const real *__restrict__ v, const emdee_lj_atom *__restrict__ atoms,
                                    const long long *__restrict__ gid, const MigRow<real> *__restrict__ rows,
                                    real *__restrict__ x2, real *__restrict__ v2, emdee_lj_atom *__restrict__ atoms2,
                                    long long *__restrict__ gid2) {
    int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n_stay + n_arrive) return;
    if (k < n_stay) {
        const int i = ids[k];
#pragma unroll
        for (int d = 0; d < 3; d++) { x2[3 * (size_t)k + d] = x[3 * (size_t)i + d]; v2[3 * (size_t)k + d] = v[3 * (size_t)i + d]; }
        atoms2[k] = atoms[i];
        gid2[k] = gid[i];
    } else {
        const MigRow<real> r = rows[k - n_stay];
#pragma unroll
        for (int d = 0; d < 3; d++) { x2[3 * (size_t)k + d] = r.x[d]; v2[3 * (size_t)k + d] = r.v[d]; }
        emdee_lj_atom a;
        a.half_sigma = r.hs; a.twice_sqrt_eps = r.te;
        atoms2[k] = a;
        gid2[k] = r.gid;
    }
}

// ------------------------------------------------------------------------------------ ghosts at a rebuild
template <typename real>
__global__ void k_dd_pack_ghost_rows(int n, const int *__restrict__ ids, const int *__restrict__ bins, DdDev<real> g,
                                     const real *__restrict__ x,
                                     const emdee_lj_atom *__restrict__ atoms, const long long *__restrict__ gid,
                                     GhostRow<real> *__restrict__ rows, int *__restrict__ codes) {
    int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    const int i = ids[k], dir = g.bin_dir[bins[k]];
    GhostRow<real> r;
#pragma unroll
    for (int d = 0; d < 3; d++) r.x[d] = x[3 * (size_t)i + d] + g.shift[dir][d];
    r.hs = atoms[i].half_sigma; r.te = atoms[i].twice_sqrt_eps;
    r.gid = gid[i];
    rows[k] = r;
    codes[k] = dir;
}

template <typename real>
__global__ void k_dd_unpack_ghost_rows(int n, const GhostRow<real> *__restrict__ rows, real *__restrict__ x,
                                       emdee_lj_atom *__restrict__ atoms, long long *__restrict__ gid) {
    int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    const GhostRow<real> r = rows[k];
#pragma unroll
    for (int d = 0; d < 3; d++) x[3 * (size_t)k + d] = r.x[d];
    emdee_lj_atom a;
    a.half_sigma = r.hs; a.twice_sqrt_eps = r.te;
    atoms[k] = a;
    gid[k] = r.gid;
}

// ------------------------------------------------------------------------------------ a rebuild in the engine's own order
// Rounds 2-4 rebuilt a domain by way of caller-order arrays: un-sort the engine's state, find the owners, assemble new
// caller arrays (stayers + arrivals), select the ghosts, append theirs, load the engine from scratch -- with the row counts
// of the two exchanges either exchanged first (round 2: two more RCCL groups, two more read-backs) or read back in between
// (round 3: capacity-padded messages, one read-back for the counts and one for the build).  Here nothing leaves the engine's
// cell order, and the host learns the counts together with the build's own words:
//   * the ownership pass reads the engine's records where they lie, strikes out the leavers and the old ghosts (keep[] = 0)
//     and selects the ghost directions of the atoms that stay in the same pass (an atom that stays does not move);
//   * leavers travel in padded messages {rows, overflow | MigRow ...} as before; the arrivals are written BEHIND the old
//     state (records, velocity planes, global ids), the rows a message did not use struck out;
//   * ghost rows likewise: out of the records, in behind the arrivals;
//   * the engine re-sorts its own slots [0, n_items) without the struck-out ones (NbSystem::resort_edit): nearly sorted
//     input, the old slot becomes the atom's id, and the number of atoms that gives is a device word until the build's
//     read-back, which carries the words below along.
// A message that overflows is seen by every rank in the same exchange (every rank neighbours every other one); the engines
// have not given up their old state by then (resort_edit keeps it in the spare buffers), so all ranks roll back and redo the
// rebuild with exact counts -- round 2's path, which the first load and the first rebuild (they fix the capacities) also take.
// Slots: [0, n) the old state | dead up to q_arr = n rounded up to PART_BLOCK | arrivals, mig capacity | ghosts, ghost capacity.
constexpr int DD_RHDR = 16;
struct DdCaps {
    int npeers;
    int start[DD_MAX_PEERS + 1];       // rows: message p holds slots [start[p], start[p+1]) of the padded buffer
    int debug_inject;                  // bounds build only (EMDEE_BOUNDS_INJECT=ghost_pack): pack ghost rows although a message overflowed
};
__host__ __device__ static inline size_t dd_pad_begin(const DdCaps &c, int p, size_t row) { return (size_t)(p + 1) * DD_RHDR + (size_t)c.start[p] * row; }
static inline size_t dd_pad_msg_begin(const DdCaps &c, int p, size_t row) { return (size_t)p * DD_RHDR + (size_t)c.start[p] * row; }
static inline size_t dd_pad_msg_bytes(const DdCaps &c, int p, size_t row) { return DD_RHDR + (size_t)(c.start[p + 1] - c.start[p]) * row; }
static inline size_t dd_pad_total(const DdCaps &c, size_t row) { return (size_t)c.npeers * DD_RHDR + (size_t)c.start[c.npeers] * row; }

__device__ __forceinline__ int dd_caps_peer(const DdCaps &c, int t) {
    int p = 0;
    while (p + 1 < c.npeers && t >= c.start[p + 1]) p++;
    return p;
}

// words a rebuild leaves for its one read-back (ints); DDW_NLIVE: the atoms of the re-sorted state (written by the engine)
constexpr int DDW_NSTAY = 0, DDW_NNEW = 1, DDW_ERR = 2, DDW_OVER = 3, DDW_NLEAVE = 4, DDW_NARRIVE = 5, DDW_NSEND = 6, DDW_NGHOST = 7,
              DDW_GSEND = 8, DDW_GRECV = 8 + DD_MAX_PEERS, DDW_ARRIVE = 8 + 2 * DD_MAX_PEERS, DDW_NLIVE = 8 + 3 * DD_MAX_PEERS,
              DDW_COUNT = 9 + 3 * DD_MAX_PEERS;

// which neighbours need an owned atom at (x, y, z) as a ghost: bit dir_bin[k] for every direction k whose halo holds it
template <typename real>
__device__ __forceinline__ unsigned dd_ghost_bits(const DdDev<real> &g, real x, real y, real z) {
    const real p[3] = {x, y, z};
    bool near_lo[3], near_hi[3];
#pragma unroll
    for (int d = 0; d < 3; d++) {
        near_lo[d] = g.cut[d] && p[d] < g.lo[d] + g.halo;
        near_hi[d] = g.cut[d] && p[d] >= g.hi[d] - g.halo;
    }
    unsigned m = 0;
    for (int k = 0; k < g.ndirs; k++) {
        bool in = true;
#pragma unroll
        for (int d = 0; d < 3; d++) {
            const int s = g.dir[k][d];
            in = in && (s == 0 || (s > 0 ? near_hi[d] : near_lo[d]));
        }
        if (in) m |= 1u << g.dir_bin[k];
    }
    return m;
}

// The ownership pass over the engine's own slots (grid: q_arr / PART_BLOCK blocks).  Owned atoms are wrapped into the global
// box (written back: a leaver's row carries the wrapped position, a stayer's does not change) and named the rank whose brick
// holds them: lmask = 1 << (1 + peer) for a leaver (bin 0 stays empty: nobody needs the list of those who stay), keep = 1
// and gmask = the ghost directions for a stayer; ghosts and the slots past n are struck out.
template <typename real>
__global__ __launch_bounds__(PART_BLOCK) void k_dd_classify_sorted(int n, int own_limit, const int *__restrict__ perm,
                                                                   Rec<real> *__restrict__ rec, DdDev<real> g,
                                                                   unsigned char *__restrict__ keep, unsigned *__restrict__ lmask,
                                                                   unsigned *__restrict__ gmask, int *__restrict__ err, int nb_leave,
                                                                   int nblk_leave, int *__restrict__ counts_leave, int nb_ghost,
                                                                   int nblk_ghost, int *__restrict__ counts_ghost) {
    const int q = blockIdx.x * PART_BLOCK + threadIdx.x;
    unsigned lm = 0, gm = 0;
    unsigned char kp = 0;
    if (q < n && perm[q] < own_limit) {
        Rec<real> r = rec[q];
        real p[3] = {r.x, r.y, r.z};
        int c[3];
        bool moved = false;
#pragma unroll
        for (int d = 0; d < 3; d++) {
            const real w = dd_wrap(p[d], g.L[d]);
            moved = moved || w != p[d];
            p[d] = w;
            c[d] = min(max((int)floor(w / g.width[d]), 0), g.grid[d] - 1);
        }
        if (moved) { r.x = p[0]; r.y = p[1]; r.z = p[2]; rec[q] = r; }
        const int dest = c[0] + g.grid[0] * (c[1] + g.grid[1] * c[2]);
        int bin = g.rank_bin[dest];
        if (bin < 0) { *err = 1; bin = 0; }
        if (bin == 0) {
            kp = 1;
            gm = dd_ghost_bits(g, p[0], p[1], p[2]);
        } else {
            lm = 1u << bin;
        }
    }
    keep[q] = kp;
    lmask[q] = lm;
    gmask[q] = gm;
    part_count_block(lm, nb_leave, nblk_leave, counts_leave);
    part_count_block(gm, nb_ghost, nblk_ghost, counts_ghost);
}

// leavers -> padded messages (ids[bin_start[1 + p] + slot], the stable partition's order), out of the engine's arrays; headers
template <typename real>
__global__ void k_dd_pack_migrants_sorted(DdCaps caps, PartView lv, const int *__restrict__ ids,
                                          const Rec<real> *__restrict__ rec, const float *__restrict__ te,
                                          const real *__restrict__ vel, size_t pitch, const long long *__restrict__ tag,
                                          unsigned char *__restrict__ buf, DdDev<real> g) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < caps.npeers) {
        int over = 0;
        for (int q = 0; q < caps.npeers; q++) over |= (lv.start(2 + q) - lv.start(1 + q)) > (caps.start[q + 1] - caps.start[q]);
        int *hdr = reinterpret_cast<int *>(buf + dd_pad_begin(caps, t, sizeof(MigRow<real>)) - DD_RHDR);
        hdr[0] = lv.start(2 + t) - lv.start(1 + t); hdr[1] = over; hdr[2] = 0; hdr[3] = 0;
    }
    if (t >= caps.start[caps.npeers]) return;
    const int p = dd_caps_peer(caps, t), slot = t - caps.start[p];
    if (slot >= lv.start(2 + p) - lv.start(1 + p)) return;
    if (!EMDEE_BOUND(BS_DD_MIG_PACK, slot, caps.start[p + 1] - caps.start[p])) return;
    const int i = ids[lv.start(1 + p) + slot];
    const Rec<real> a = rec[i];
    MigRow<real> r;
    r.x[0] = a.x - g.mig_off[1 + p][0]; r.x[1] = a.y - g.mig_off[1 + p][1]; r.x[2] = a.z - g.mig_off[1 + p][2];
#pragma unroll
    for (int d = 0; d < 3; d++) r.v[d] = vel[d * pitch + i];
    rec_params(rec, te, i, r.hs, r.te);
    r.gid = tag[i];
    reinterpret_cast<MigRow<real> *>(buf + dd_pad_begin(caps, p, sizeof(MigRow<real>)))[slot] = r;
}

// After the migrant exchange (grid: blocks of PART_BLOCK over the migrant capacity, at least one): arrival t of the padded
// receive buffer -> slot q_arr + t behind the old state (record, velocities, global id, ghost directions), unused rows struck
// out; thread 0: the counts of the migration and the overflow word (mine or anybody's) for the read-back.
template <typename real>
__global__ __launch_bounds__(PART_BLOCK) void k_dd_unpack_arrivals(DdCaps caps, PartView lv, const int *__restrict__ err,
                                                                   const unsigned char *__restrict__ recv, int n_owned_old, int q_arr,
                                                                   Rec<real> *__restrict__ rec, float *__restrict__ te,
                                                                   real *__restrict__ vel, size_t pitch, long long *__restrict__ tag,
                                                                   unsigned char *__restrict__ keep, unsigned *__restrict__ gmask,
                                                                   DdDev<real> g, int nb_ghost, int nblk_ghost,
                                                                   int *__restrict__ counts_ghost, int blk0, int *__restrict__ w) {
    const int t = blockIdx.x * PART_BLOCK + threadIdx.x;
    if (t == 0) {
        int over = 0, arrive = 0, leave = 0;
        for (int p = 0; p < caps.npeers; p++) {
            const int cap = caps.start[p + 1] - caps.start[p];
            const int *hdr = reinterpret_cast<const int *>(recv + dd_pad_begin(caps, p, sizeof(MigRow<real>)) - DD_RHDR);
            const int out = lv.start(2 + p) - lv.start(1 + p);
            over |= hdr[1] | (hdr[0] > cap) | (out > cap);
            const int in = min(max(hdr[0], 0), cap);
            w[DDW_ARRIVE + p] = in;
            arrive += in; leave += out;
        }
        w[DDW_NSTAY] = n_owned_old - leave;
        w[DDW_NNEW] = n_owned_old - leave + arrive;
        w[DDW_ERR] = *err;
        w[DDW_OVER] = over;
        w[DDW_NLEAVE] = leave;
        w[DDW_NARRIVE] = arrive;
    }
    unsigned gm = 0;
    if (t < caps.start[caps.npeers]) {
        const int p = dd_caps_peer(caps, t), slot = t - caps.start[p], cap = caps.start[p + 1] - caps.start[p];
        const int *hdr = reinterpret_cast<const int *>(recv + dd_pad_begin(caps, p, sizeof(MigRow<real>)) - DD_RHDR);
        const int q = q_arr + t;
        unsigned char kp = 0;
        if (slot < min(max(hdr[0], 0), cap)) {
            const MigRow<real> r = reinterpret_cast<const MigRow<real> *>(recv + dd_pad_begin(caps, p, sizeof(MigRow<real>)))[slot];
            store_rec<real>(rec, te, q, r.x[0], r.x[1], r.x[2], r.hs, r.te);
#pragma unroll
            for (int d = 0; d < 3; d++) vel[d * pitch + q] = r.v[d];
            tag[q] = r.gid;
            gm = dd_ghost_bits(g, r.x[0], r.x[1], r.x[2]);
            kp = 1;
        }
        keep[q] = kp;
    }
    // (the ghost masks of the arrival blocks exist for every thread of the grid: the scatter reads whole blocks)
    gmask[q_arr + t] = gm;
    part_count_block(gm, nb_ghost, nblk_ghost, counts_ghost + blk0);
}

// ghost rows -> padded messages, out of the engine's arrays; peer_count[p] = entries bound for peer p (k_part_starts), list
// entry k of peer p sits at sum of the counts before p + slot; codes[k] = direction, for the per-step messages
template <typename real>
__global__ void k_dd_pack_ghost_rows_sorted(DdCaps caps, PartView gv, DdBins pb, const int *__restrict__ ids,
                                            const int *__restrict__ bins, DdDev<real> g, const Rec<real> *__restrict__ rec,
                                            const float *__restrict__ te, const long long *__restrict__ tag,
                                            unsigned char *__restrict__ buf, int *__restrict__ codes, const int *__restrict__ w_over) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    auto peer_count = [&](int q) { return gv.start(pb.lo[q + 1]) - gv.start(pb.lo[q]); };   // entries bound for peer q
    int over = w_over[DDW_OVER];
    for (int q = 0; q < caps.npeers; q++) over |= peer_count(q) > (caps.start[q + 1] - caps.start[q]);
    if (t < caps.npeers) {
        int *hdr = reinterpret_cast<int *>(buf + dd_pad_begin(caps, t, sizeof(GhostRow<real>)) - DD_RHDR);
        // hdr[2]: my error word (an atom that left the neighbourhood of its brick) -- every rank neighbours every other one, so
        // all of them see it in this exchange and fail together, instead of one throwing while the others walk into the next
        // step's send / receive with a peer that is gone
        hdr[0] = peer_count(t); hdr[1] = over; hdr[2] = w_over[DDW_ERR]; hdr[3] = 0;
    }
    // an overflow anywhere: the rebuild will be redone with counts and no row of this one is looked at -- and the send
    // list (ids, bins, codes: sized by the capacities) does not hold what the counts say
    // (bounds build, test of the checker itself: debug_inject puts the pre-2d85cf1 behaviour back -- the list indexed by the counts)
    if ((over && !caps.debug_inject) || t >= caps.start[caps.npeers]) return;
    const int p = dd_caps_peer(caps, t), slot = t - caps.start[p];
    if (slot >= peer_count(p)) return;
    int k = slot;
    for (int q = 0; q < p; q++) k += peer_count(q);
    if (!EMDEE_BOUND(BS_DD_GHOST_PACK, k, caps.start[caps.npeers])) return;   // the send list (ids, bins, codes) holds start[npeers] entries
    const int i = ids[k], dir = g.bin_dir[bins[k]];
    const Rec<real> a = rec[i];
    GhostRow<real> r;
    r.x[0] = a.x + g.shift[dir][0]; r.x[1] = a.y + g.shift[dir][1]; r.x[2] = a.z + g.shift[dir][2];
    rec_params(rec, te, i, r.hs, r.te);
    r.gid = tag[i];
    reinterpret_cast<GhostRow<real> *>(buf + dd_pad_begin(caps, p, sizeof(GhostRow<real>)))[slot] = r;
    codes[k] = dir;
}

// After the ghost exchange (grid: blocks over the ghost capacity, at least one): row t of the padded receive buffer -> slot
// ghost_base + t (record, global id), unused rows struck out; thread 0: send / receive counts per peer, totals, the overflow
// and error words of either exchange (mine or a peer's: the same words on every rank).
template <typename real>
__global__ void k_dd_unpack_ghost_rows_sorted(DdCaps scaps, DdCaps rcaps, PartView gv, DdBins pb,
                                              const unsigned char *__restrict__ recv, int ghost_base, Rec<real> *__restrict__ rec,
                                              float *__restrict__ te, real *__restrict__ vel, size_t pitch,
                                              long long *__restrict__ tag, unsigned char *__restrict__ keep, int *__restrict__ w) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t == 0) {
        int over = w[DDW_OVER], nsend = 0, nghost = 0, err = w[DDW_ERR];
        for (int p = 0; p < scaps.npeers; p++) {
            const int rcap = rcaps.start[p + 1] - rcaps.start[p], scap = scaps.start[p + 1] - scaps.start[p];
            const int *hdr = reinterpret_cast<const int *>(recv + dd_pad_begin(rcaps, p, sizeof(GhostRow<real>)) - DD_RHDR);
            const int mine = gv.start(pb.lo[p + 1]) - gv.start(pb.lo[p]);      // entries bound for peer p
            over |= hdr[1] | (hdr[0] > rcap) | (mine > scap);
            err |= hdr[2];
            w[DDW_GSEND + p] = mine;
            w[DDW_GRECV + p] = hdr[0];
            nsend += mine;
            nghost += hdr[0];
        }
        w[DDW_OVER] = over;
        w[DDW_ERR] = err;
        w[DDW_NSEND] = nsend;
        w[DDW_NGHOST] = nghost;
    }
    if (t >= rcaps.start[rcaps.npeers]) return;
    const int p = dd_caps_peer(rcaps, t), slot = t - rcaps.start[p], cap = rcaps.start[p + 1] - rcaps.start[p];
    const int *hdr = reinterpret_cast<const int *>(recv + dd_pad_begin(rcaps, p, sizeof(GhostRow<real>)) - DD_RHDR);
    const int q = ghost_base + t;
    // (a sender that overflowed packed no row at all: its count says more than its message holds -- nothing of it is kept)
    const bool have = hdr[1] == 0 && slot < min(max(hdr[0], 0), cap);
    if (have) {
        const GhostRow<real> r = reinterpret_cast<const GhostRow<real> *>(recv + dd_pad_begin(rcaps, p, sizeof(GhostRow<real>)))[slot];
        if (EMDEE_BOUND(BS_DD_GHOST_UNPACK, slot, cap)) {
            store_rec<real>(rec, te, q, r.x[0], r.x[1], r.x[2], r.hs, r.te);
            // (a ghost in a cell that also holds owned atoms goes through the owner lane of the step kernel like they do -- no
            // row, no force: with zero velocity it stays where the halo put it and never trips the rebuild trigger)
#pragma unroll
            for (int d = 0; d < 3; d++) vel[d * pitch + q] = (real)0;
            tag[q] = r.gid;
        }
    }
    keep[q] = have ? 1 : 0;
}

// ------------------------------------------------------------------------------------ in-process transport
// Validation mode (all domains of the grid in one process on one device): a receiving domain pulls the messages of all
// its peers with ONE launch -- segment blockIdx.y of `a`, 4-byte words (message offsets and sizes are multiples of 4) --
// where round 2 queued one device copy per message: 56 copies per step of an 8-domain grid, and the one host thread that
// drives all eight domains was what the rehearsal measured.
struct PullSeg {
    const unsigned *src;
    unsigned *dst;
    unsigned words;
};
struct PullArgs {
    int n;
    PullSeg seg[DD_MAX_PEERS];
};
static __global__ void k_dd_pull(PullArgs a) {
    const PullSeg s = a.seg[blockIdx.y];
    for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < s.words; i += gridDim.x * blockDim.x) s.dst[i] = s.src[i];
}

// ------------------------------------------------------------------------------------ per-step halo messages
// Message to peer p: [int request | pad to DD_HDR][3 reals per atom].  The request word is this domain's
// "one of my atoms has moved skin/2" flag for the positions being sent: it rides on the halo message, so the
// rebuild decision of a step needs no collective of its own (with at most 3 bricks per dimension every rank is
// a neighbour of every other one, and the messages of one step are an all-gather of the flags).
template <typename real>
__global__ void k_dd_pack_step(int n, DdPlan plan, const int *__restrict__ ids, const int *__restrict__ codes,
                               DdDev<real> g, const int *__restrict__ inv_perm, const Rec<real> *__restrict__ rec,
                               const int *__restrict__ my_flag, unsigned char *__restrict__ buf) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < plan.npeers) {
        int *hdr = reinterpret_cast<int *>(buf + (size_t)k * DD_HDR + (size_t)plan.send_start[k] * 3 * sizeof(real));
        hdr[0] = *my_flag; hdr[1] = 0; hdr[2] = 0; hdr[3] = 0;
    }
    if (k >= n) return;
    int p = 0;
    while (p + 1 < plan.npeers && k >= plan.send_start[p + 1]) p++;
    const Rec<real> r = rec[inv_perm[ids[k]]];
    const int c = codes[k];
    real *out = reinterpret_cast<real *>(buf + (size_t)(p + 1) * DD_HDR) + 3 * (size_t)k;
    out[0] = r.x + g.shift[c][0]; out[1] = r.y + g.shift[c][1]; out[2] = r.z + g.shift[c][2];
}

// ghost k of the received messages -> its record (cell order); *global_flag |= every sender's request | mine
template <typename real>
__global__ void k_dd_unpack_step(int n, int n_owned, DdPlan plan, const int *__restrict__ inv_perm,
                                 const unsigned char *__restrict__ buf, Rec<real> *__restrict__ rec,
                                 const int *__restrict__ my_flag, int *__restrict__ global_flag) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < plan.npeers) {
        const int *hdr = reinterpret_cast<const int *>(buf + (size_t)k * DD_HDR + (size_t)plan.recv_start[k] * 3 * sizeof(real));
        if (hdr[0] != 0) *global_flag = 1;
    }
    if (k == 0 && *my_flag != 0) *global_flag = 1;
    if (k >= n) return;
    int p = 0;
    while (p + 1 < plan.npeers && k >= plan.recv_start[p + 1]) p++;
    const real *in = reinterpret_cast<const real *>(buf + (size_t)(p + 1) * DD_HDR) + 3 * (size_t)k;
    const int q = inv_perm[plan.ghost_id[p] + (k - plan.recv_start[p])];
    Rec<real> r = rec[q];
    r.x = in[0]; r.y = in[1]; r.z = in[2];
    rec[q] = r;
}

// start of a batch of queued steps: words[0] keeps the request raised for the current positions, the rest is cleared
static __global__ void k_dd_batch_begin(int *__restrict__ words, int nwords, int carry) {
    const int t = threadIdx.x;
    const int keep = words[carry];
    __syncthreads();
    if (t < nwords) words[t] = (t == 0) ? keep : 0;
}

}  // namespace emdee
