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
};

// rows that travel at a rebuild
template <typename real>
struct MigRow {                       // an atom changing owner
    real x[3], v[3];
    float hs, te;
    long long gid;
};
template <typename real>
struct GhostRow {                     // a ghost as first sent: shifted position + LJAtom
    real x[3];
    float hs, te;
};

// prefix of atoms per peer for the per-step messages (message p = DD_HDR bytes + 3 reals per atom)
struct DdPlan {
    int npeers;
    int send_start[DD_MAX_PEERS + 1];
    int recv_start[DD_MAX_PEERS + 1];
};
// message p begins at byte p DD_HDR + 3 w start[p] and holds DD_HDR + 3 w (start[p+1] - start[p]) bytes
static inline size_t dd_msg_begin(const int *start, int p, size_t w) { return (size_t)p * DD_HDR + (size_t)start[p] * 3 * w; }
static inline size_t dd_msg_bytes(const int *start, int p, size_t w) { return DD_HDR + (size_t)(start[p + 1] - start[p]) * 3 * w; }

// ------------------------------------------------------------------------------------ stable multi-bin partition
// Items carry a bit mask of bins (<= 32 bins; an item may sit in several: a corner atom is a ghost of up to
// seven neighbours).  Output: for every bin, the ids of its items in ascending order -- deterministic, no
// atomics on the output.  count -> exclusive scan of counts[bin][block] -> scatter.  The kernels that produce the
// masks (ownership, ghost selection) count their block themselves: blocks of PART_BLOCK items.
constexpr int PART_BLOCK = 256;

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
            real p = x[3 * (size_t)i + d];
            p -= g.L[d] * floor(p / g.L[d]);
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
struct DdBins {
    int npeers;
    int lo[DD_MAX_PEERS + 2];
};
static __global__ void k_part_starts(int nbins, int nblocks, const int *__restrict__ offs, int *__restrict__ bin_start,
                                     DdBins pb, int *__restrict__ peer_count) {
    const int t = threadIdx.x;
    if (t <= nbins) bin_start[t] = offs[(size_t)t * nblocks];
    if (t < pb.npeers) peer_count[t] = offs[(size_t)pb.lo[t + 1] * nblocks] - offs[(size_t)pb.lo[t] * nblocks];
}

// ------------------------------------------------------------------------------------ migration rows
template <typename real>
__global__ void k_dd_pack_migrants(int first, int total, const int *__restrict__ ids, const real *__restrict__ x,
                                   const real *__restrict__ v, const emdee_lj_atom *__restrict__ atoms,
                                   const long long *__restrict__ gid, MigRow<real> *__restrict__ rows) {
    int k = first + blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= total) return;
    const int i = ids[k];
    MigRow<real> r;
#pragma unroll
    for (int d = 0; d < 3; d++) { r.x[d] = x[3 * (size_t)i + d]; r.v[d] = v[3 * (size_t)i + d]; }
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
                                     const emdee_lj_atom *__restrict__ atoms, GhostRow<real> *__restrict__ rows,
                                     int *__restrict__ codes) {
    int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    const int i = ids[k], dir = g.bin_dir[bins[k]];
    GhostRow<real> r;
#pragma unroll
    for (int d = 0; d < 3; d++) r.x[d] = x[3 * (size_t)i + d] + g.shift[dir][d];
    r.hs = atoms[i].half_sigma; r.te = atoms[i].twice_sqrt_eps;
    rows[k] = r;
    codes[k] = dir;
}

template <typename real>
__global__ void k_dd_unpack_ghost_rows(int n, const GhostRow<real> *__restrict__ rows, real *__restrict__ x,
                                       emdee_lj_atom *__restrict__ atoms) {
    int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    const GhostRow<real> r = rows[k];
#pragma unroll
    for (int d = 0; d < 3; d++) x[3 * (size_t)k + d] = r.x[d];
    emdee_lj_atom a;
    a.half_sigma = r.hs; a.twice_sqrt_eps = r.te;
    atoms[k] = a;
}

// ------------------------------------------------------------------------------------ count-free rebuild messages
// Round 2 exchanged the row counts of a migration and of a ghost selection first (two extra RCCL groups and two blocking
// read-backs per rebuild) because the counts size the messages.  Here a message has a CAPACITY both ends know without
// talking -- migrants: a fixed number of rows per peer; ghost rows: the count of the previous rebuild plus an eighth --
// and starts with a 16-byte header {rows, overflow, 0, 0}; everything between the two exchanges runs on device-side
// counts over upper-bound grids, and ONE read-back at the end tells the host all of them.  "overflow" is raised in ALL
// messages of a rank whose rows for some peer exceed the capacity; every rank neighbours every other one, so everybody
// learns of it in the same exchange and the whole rebuild is redone with exact counts (dd.hpp).
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

// words a count-free rebuild leaves for its one read-back (ints)
constexpr int DDW_NSTAY = 0, DDW_NNEW = 1, DDW_ERR = 2, DDW_OVER = 3, DDW_NLEAVE = 4, DDW_NARRIVE = 5, DDW_NSEND = 6, DDW_NGHOST = 7,
              DDW_GSEND = 8, DDW_GRECV = 8 + DD_MAX_PEERS, DDW_ARRIVE = 8 + 2 * DD_MAX_PEERS, DDW_COUNT = 8 + 3 * DD_MAX_PEERS;

// leavers -> padded messages (ids[bin_start[1 + p] + slot], the stable partition's order); headers
template <typename real>
__global__ void k_dd_pack_migrants_padded(DdCaps caps, const int *__restrict__ bin_start, const int *__restrict__ ids,
                                          const real *__restrict__ x, const real *__restrict__ v,
                                          const emdee_lj_atom *__restrict__ atoms, const long long *__restrict__ gid,
                                          unsigned char *__restrict__ buf) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < caps.npeers) {
        int over = 0;
        for (int q = 0; q < caps.npeers; q++) over |= (bin_start[2 + q] - bin_start[1 + q]) > (caps.start[q + 1] - caps.start[q]);
        int *hdr = reinterpret_cast<int *>(buf + dd_pad_begin(caps, t, sizeof(MigRow<real>)) - DD_RHDR);
        hdr[0] = bin_start[2 + t] - bin_start[1 + t]; hdr[1] = over; hdr[2] = 0; hdr[3] = 0;
    }
    if (t >= caps.start[caps.npeers]) return;
    const int p = dd_caps_peer(caps, t), slot = t - caps.start[p];
    if (slot >= bin_start[2 + p] - bin_start[1 + p]) return;
    if (!EMDEE_BOUND(BS_DD_MIG_PACK, slot, caps.start[p + 1] - caps.start[p])) return;
    const int i = ids[bin_start[1 + p] + slot];
    MigRow<real> r;
#pragma unroll
    for (int d = 0; d < 3; d++) { r.x[d] = x[3 * (size_t)i + d]; r.v[d] = v[3 * (size_t)i + d]; }
    r.hs = atoms[i].half_sigma; r.te = atoms[i].twice_sqrt_eps;
    r.gid = gid[i];
    reinterpret_cast<MigRow<real> *>(buf + dd_pad_begin(caps, p, sizeof(MigRow<real>)))[slot] = r;
}

// after the migrant exchange: w[DDW_NSTAY], w[DDW_NNEW], arrivals per peer, overflow (mine or anybody's), error word
template <typename real>
__global__ void k_dd_migrant_counts(DdCaps caps, const int *__restrict__ bin_start, const int *__restrict__ err,
                                    const unsigned char *__restrict__ recv, int *__restrict__ w) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    int over = 0, arrive = 0, leave = 0;
    for (int p = 0; p < caps.npeers; p++) {
        const int cap = caps.start[p + 1] - caps.start[p];
        const int *hdr = reinterpret_cast<const int *>(recv + dd_pad_begin(caps, p, sizeof(MigRow<real>)) - DD_RHDR);
        const int out = bin_start[2 + p] - bin_start[1 + p];
        over |= hdr[1] | (hdr[0] > cap) | (out > cap);
        const int in = min(max(hdr[0], 0), cap);
        w[DDW_ARRIVE + p] = in;
        arrive += in; leave += out;
    }
    w[DDW_NSTAY] = bin_start[1];
    w[DDW_NNEW] = bin_start[1] + arrive;
    w[DDW_ERR] = *err;
    w[DDW_OVER] = over;
    w[DDW_NLEAVE] = leave;
    w[DDW_NARRIVE] = arrive;
}

// new owned arrays from device-side counts: stayers (ids[0 .. n_stay)), then the arrivals in peer order
template <typename real>
__global__ void k_dd_assemble_padded(int n_max, DdCaps caps, const int *__restrict__ w, const int *__restrict__ ids,
                                     const real *__restrict__ x, const real *__restrict__ v, const emdee_lj_atom *__restrict__ atoms,
                                     const long long *__restrict__ gid, const unsigned char *__restrict__ recv,
                                     real *__restrict__ x2, real *__restrict__ v2, emdee_lj_atom *__restrict__ atoms2,
                                     long long *__restrict__ gid2) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    const int n_stay = w[DDW_NSTAY], n_new = min(w[DDW_NNEW], n_max);
    if (k >= n_new) return;
    if (k < n_stay) {
        const int i = ids[k];
#pragma unroll
        for (int d = 0; d < 3; d++) { x2[3 * (size_t)k + d] = x[3 * (size_t)i + d]; v2[3 * (size_t)k + d] = v[3 * (size_t)i + d]; }
        atoms2[k] = atoms[i];
        gid2[k] = gid[i];
        return;
    }
    int a = k - n_stay, p = 0;
    while (p + 1 < caps.npeers && a >= w[DDW_ARRIVE + p]) { a -= w[DDW_ARRIVE + p]; p++; }
    if (!EMDEE_BOUND(BS_DD_ASSEMBLE, k, n_max)) return;
    const MigRow<real> r = reinterpret_cast<const MigRow<real> *>(recv + dd_pad_begin(caps, p, sizeof(MigRow<real>)))[a];
#pragma unroll
    for (int d = 0; d < 3; d++) { x2[3 * (size_t)k + d] = r.x[d]; v2[3 * (size_t)k + d] = r.v[d]; }
    emdee_lj_atom at;
    at.half_sigma = r.hs; at.twice_sqrt_eps = r.te;
    atoms2[k] = at;
    gid2[k] = r.gid;
}

// ghost rows -> padded messages; peer_count[p] = entries bound for peer p (k_part_starts), list entry k of peer p sits at
// sum of the counts before p + slot; codes[k] = direction, for the per-step messages
template <typename real>
__global__ void k_dd_pack_ghost_rows_padded(DdCaps caps, const int *__restrict__ peer_count, const int *__restrict__ ids,
                                            const int *__restrict__ bins, DdDev<real> g, const real *__restrict__ x,
                                            const emdee_lj_atom *__restrict__ atoms, unsigned char *__restrict__ buf,
                                            int *__restrict__ codes, const int *__restrict__ w_over) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    int over = w_over[DDW_OVER];
    for (int q = 0; q < caps.npeers; q++) over |= peer_count[q] > (caps.start[q + 1] - caps.start[q]);
    if (t < caps.npeers) {
        int *hdr = reinterpret_cast<int *>(buf + dd_pad_begin(caps, t, sizeof(GhostRow<real>)) - DD_RHDR);
        // hdr[2]: my error word (an atom that left the neighbourhood of its brick) -- every rank neighbours every other one, so
        // all of them see it in this exchange and fail together, instead of one throwing while the others walk into the next
        // step's send / receive with a peer that is gone
        hdr[0] = peer_count[t]; hdr[1] = over; hdr[2] = w_over[DDW_ERR]; hdr[3] = 0;
    }
    // an overflow anywhere: the rebuild will be redone with counts and no row of this one is looked at -- and the send
    // list (ids, bins, codes: sized by the capacities) does not hold what the counts say
    // (bounds build, test of the checker itself: debug_inject puts the pre-2d85cf1 behaviour back -- the list indexed by the counts)
    if ((over && !caps.debug_inject) || t >= caps.start[caps.npeers]) return;
    const int p = dd_caps_peer(caps, t), slot = t - caps.start[p];
    if (slot >= peer_count[p]) return;
    int k = slot;
    for (int q = 0; q < p; q++) k += peer_count[q];
    if (!EMDEE_BOUND(BS_DD_GHOST_PACK, k, caps.start[caps.npeers])) return;   // the send list (ids, bins, codes) holds start[npeers] entries
    const int i = ids[k], dir = g.bin_dir[bins[k]];
    GhostRow<real> r;
#pragma unroll
    for (int d = 0; d < 3; d++) r.x[d] = x[3 * (size_t)i + d] + g.shift[dir][d];
    r.hs = atoms[i].half_sigma; r.te = atoms[i].twice_sqrt_eps;
    reinterpret_cast<GhostRow<real> *>(buf + dd_pad_begin(caps, p, sizeof(GhostRow<real>)))[slot] = r;
    codes[k] = dir;
}

// after the ghost exchange: send / receive counts per peer, totals, overflow of either exchange
template <typename real>
__global__ void k_dd_ghost_counts(DdCaps scaps, DdCaps rcaps, const int *__restrict__ peer_count,
                                  const unsigned char *__restrict__ recv, int *__restrict__ w) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    int over = w[DDW_OVER], nsend = 0, nghost = 0, err = w[DDW_ERR];
    for (int p = 0; p < scaps.npeers; p++) {
        const int rcap = rcaps.start[p + 1] - rcaps.start[p], scap = scaps.start[p + 1] - scaps.start[p];
        const int *hdr = reinterpret_cast<const int *>(recv + dd_pad_begin(rcaps, p, sizeof(GhostRow<real>)) - DD_RHDR);
        over |= hdr[1] | (hdr[0] > rcap) | (peer_count[p] > scap);
        err |= hdr[2];
        w[DDW_GSEND + p] = peer_count[p];
        w[DDW_GRECV + p] = hdr[0];
        nsend += peer_count[p];
        nghost += hdr[0];
    }
    w[DDW_OVER] = over;
    w[DDW_ERR] = err;                  // mine or a peer's: the same word on every rank
    w[DDW_NSEND] = nsend;
    w[DDW_NGHOST] = nghost;
}

// padded ghost messages -> the ghosts behind the owned atoms (recv_start: prefix of the receive counts, now known)
template <typename real>
__global__ void k_dd_unpack_ghost_rows_padded(int n, DdPlan plan, DdCaps rcaps, const unsigned char *__restrict__ recv,
                                              real *__restrict__ x, emdee_lj_atom *__restrict__ atoms) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    int p = 0;
    while (p + 1 < plan.npeers && k >= plan.recv_start[p + 1]) p++;
    const GhostRow<real> r = reinterpret_cast<const GhostRow<real> *>(recv + dd_pad_begin(rcaps, p, sizeof(GhostRow<real>)))[k - plan.recv_start[p]];
#pragma unroll
    for (int d = 0; d < 3; d++) x[3 * (size_t)k + d] = r.x[d];
    emdee_lj_atom a;
    a.half_sigma = r.hs; a.twice_sqrt_eps = r.te;
    atoms[k] = a;
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
    const int q = inv_perm[n_owned + k];
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
