// brick.hpp -- LDS-tiled neighbour build and force kernels (the production path).
//
// Why: in the direct kernels of kernels.hpp every neighbour costs two global line look-ups
// (one 32-B record = 2 x dwordx4), and the per-CU texture/L1 path, not HBM and not the fp64
// VALU, sets the time (profiles/r01: 4.84 ms per 10^7-atom force pass).  Here a workgroup owns
// a BRICK of BX x BY x BZ cells and first stages the brick plus its one-cell halo -- the TILE,
// (BX+2)(BY+2)(BZ+2) cells -- from HBM into LDS with unit-stride, fully coalesced record
// loads.  Periodic images are resolved while staging (a wrapped cell gets its +-L shift added
// once), so the pair loop has no minimum-image arithmetic at all.  Neighbour lists then hold
// 16-bit TILE-LOCAL slots: half the index bytes of a global int32 list, and every gather is an
// LDS read.  The tile of a brick is a pure function of the cell populations, which are frozen
// between rebuilds, so slots written by the build kernel stay valid for every force pass.
//
// Work split inside a workgroup: groups of G lanes share one atom (G = 4 in the default variant: 16 atoms
// per wavefront, so that the work outside the pair loop is shared by many atoms; 8 or 16 in the tuning
// variants).  Lane l of the group takes neighbours l, l+G, l+2G, ...; the row is stored LANE-MAJOR in blocks
// of 8 G entries so that those are 8 consecutive uint16 = ONE 16-byte load per lane, issued one atom ahead
// of the arithmetic.  Per-lane partial sums are combined with DPP row shifts.  Owner-computes, full list: no
// atomics, no pre-zeroing.
//
// k_brick_build keeps its tile as fp32 positions relative to the brick origin (16 B per record:
// twice the occupancy of the fp64 tile) and tests r^2 < r_list^2 in fp32 first.  The fp32 result is
// trusted only outside a rounding band whose width is a proven bound of the fp32 error; the few
// pairs inside the band are re-tested with the exact fp64 records, so the neighbour set is
// bit-identical to an all-fp64 build.
//
// Reference lines restated: pair function src/lennard_jones.jl:25-42 (lj_pair.hpp);
// f_ij = W/r2 * r_ij and the half split of E, W per atom src/nonbonded.jl:136-145; the cell
// grid convention src/cells.jl:82-85.  What replaces what: compute_tile! (src/nonbonded.jl:44-107)
// and find_action_partners1! (src/cells.jl:224-297).
#pragma once

#include "kernels.hpp"

namespace emdee {

// BRICK_STEP = the force pass with the velocity-Verlet update fused in: the owner lane of each atom adds
// v += c f / m, x += dt v straight from its registers and writes the NEXT position buffer (positions are
// ping-ponged, other workgroups are still staging the current ones), so an inner MD step is one kernel:
// no separate kick/drift pass and no force array round trip.
enum BrickMode { BRICK_FORCE = 1, BRICK_STATS = 2, BRICK_STEP = 3 };

// a loop the compiler must leave as written (no unrolling, no interleaving of trips behind run-time alias checks)
#ifndef EMDEE_NO_PLAIN_LOOPS
#define EMDEE_PLAIN_LOOP _Pragma("clang loop unroll(disable) vectorize(disable) interleave(disable)")
#else
#define EMDEE_PLAIN_LOOP
#endif

constexpr int EPL = 8;   // neighbour entries per lane per 16-byte load
// entries the force kernels read of every row without looking at its length: the NPF prefetched blocks of 8 G entries
// and the one after them (rows are sentinel-padded, so the row stride must hold them)
// (1024-thread workgroups with 4 lanes per atom are the long-row variant -- rc = 3.5 sigma: 184 entries -- and prefetch five
// 32-entry blocks: with two, every further block of a row is a dependent global load that four wavefronts per SIMD cannot hide)
constexpr int brick_prefetch_blocks(int G, int THREADS) { return (EPL * G) >= 128 ? 1 : ((G == 4 && THREADS >= 1024) ? 5 : 2); }
constexpr int brick_min_stride(int G, int THREADS) { return (brick_prefetch_blocks(G, THREADS) + 1) * EPL * G; }

template <int BX_, int BY_, int BZ_>
struct BrickShape {
    static constexpr int BX = BX_, BY = BY_, BZ = BZ_;
    static constexpr int TX = BX + 2, TY = BY + 2, TZ = BZ + 2;
    static constexpr int NTC = TX * TY * TZ;   // tile cells
    static constexpr int NOC = BX * BY * BZ;   // own cells
};

struct BrickGrid {
    int nb[3];         // bricks per dimension
    int nbricks;
    int per_xcd;       // ceil(nbricks / 8): XCD-contiguous remap of block ids
    // interior sub-box (bricks whose tile holds no ghost cell): phase-1 launches enumerate only these,
    // so that every XCD gets an equal share of them
    int ib_lo[3], ib_n[3];
    int ib_count, ib_per_xcd;
    // boundary bricks = the rest, enumerated densely as z-slabs, then y-slabs, then x-slabs (phase 2)
    int bb_z, bb_y, bb_count, bb_per_xcd;   // bb_z = #bricks in the z-slabs, bb_y = # in the y-slabs
};

// k-th index of [0, nb) outside the interval [lo, lo + n)
__host__ __device__ __forceinline__ int outside_interval(int k, int lo, int n) { return k < lo ? k : k + n; }

// tile record in LDS: same bytes as the HBM record (fp64 32 B, fp32 16 B + te plane)
template <typename real>
__device__ __forceinline__ void tile_load(const Rec<real> *tile, const float *tile_te, int s, real &x, real &y, real &z,
                                          real &hs, real &te);
template <>
__device__ __forceinline__ void tile_load<double>(const Rec<double> *tile, const float *, int s, double &x, double &y,
                                                  double &z, double &hs, double &te) {
    Rec<double> r = tile[s];   // 2 x ds_read_b128
    x = r.x; y = r.y; z = r.z; hs = (double)r.hs; te = (double)r.te;
}
template <>
__device__ __forceinline__ void tile_load<float>(const Rec<float> *tile, const float *tile_te, int s, float &x, float &y,
                                                 float &z, float &hs, float &te) {
    Rec<float> r = tile[s];    // ds_read_b128
    x = r.x; y = r.y; z = r.z; hs = r.hs; te = tile_te[s];
}

template <typename real>
struct BrickArgs {
    int n, n_owned;            // slots in use; ids below n_owned are owned atoms, the others ghosts (ids may have gaps)
    int any_ghosts;
    const Rec<real> *rec;
    const float *te;
    const int *perm;
    const int *start;          // cell -> first slot (cell order)
    GridP<real> g;
    BrickGrid bg;
    int tile_cap;              // records the dynamic LDS tile can hold
    int own_cap;               // own atoms the per-atom LDS table can hold
    unsigned short *nbr;       // ELL rows of tile-local slots, lane-major blocks (see row_position)
    int stride;                // entries per row, a multiple of 8 G
    int *cnt;
    int *flags;                // [0] row overflow (max count), [2] tile overflow
    real rlist2;
    float margin;              // build: half-width of the fp32 rounding band around rlist2
    // near/far build (ALG 23): tile coordinates are scaled by nf_scale = k so that k^2 (r_near^2 - r_list^2) = -2.0 exactly --
    // the class of a candidate is then the top two bits of the float k^2 (d^2 - r_list^2): sign = listed, sign and |.| >= 2 = near
    float nf_scale, nf_scale2;
    LJModel<real> model;
    size_t pitch;
    real *frc, *en, *vir;
    unsigned long long *stats; // BRICK_STATS: [0] entries, [1] max row, [2] in-cutoff entries
    int phase;                 // 0: every brick; 1: bricks whose tile has no ghost cell; 2: the others
    // BRICK_STEP only
    real *vel;                 // SoA planes of the current velocities (read)
    real *vel_next;            // ... and of the next ones (written): velocities ping-pong with the positions, so a
                               // step whose results must be discarded (decomposed runs: a neighbour asked for a
                               // rebuild while the interior bricks were already integrating) leaves no trace
    const real *xb;            // positions at the last build (rebuild trigger)
    const real *inv_mass;      // may be NULL
    Rec<real> *rec_next;       // position buffer of the next step
    real kick_c, dt, thr2;
    int *trigger;
    const int *guard;          // run-ahead launches: do nothing if the previous step asked for a rebuild
    int *btab;                 // per-brick tables written once per rebuild (k_brick_tables); NULL = compute them here
    // BRICK_FORCE, operator path: results straight into the caller's arrays (caller order, 3 x N / N), no unsort pass
    real *user_f, *user_e, *user_w;
    const real *noise;         // Langevin O step between kick and drift: v = lgv_c1 v + noise[p]; NULL = NVE
    real lgv_c1;
    // UNI kernels: every atom carries the same LJAtom, so sigma_ij^2 and 4 eps_ij are launch constants
    real uni_sigma2, uni_e4;
    LJUni<real> uni;           // ... folded into the switch constants for the force-only kernels of the MD loop (lj_pair.hpp)
    // list entries are tile slots shifted left by idx_shift: single-species boxes store BYTE offsets into the coordinate
    // planes (slot * sizeof(real); a tile of <= 2048 slots keeps them below 2^16), saving the address shift per pair
    int idx_shift;
    // fp32 operator path only (BRICK_FORCE, record tile): the reference's own Float32 arithmetic -- the tile holds scaled
    // positions s = x / L and every pair takes L (ds - round(ds)) (src/nonbonded.jl:40,52-61,70) instead of staged images
    int refmath;
    // typed boxes (two species, typed.hpp): per-(cell, species) starts and the pair constants sigma_ij^2, 4 eps_ij, [ti * 2 + tj]
    const int *tstart;
    int tdig = 1;    // words of tstart per (cell, species) block: 4 when a two-species box is sorted by x quarter as well
    real tsig2[4], te4[4];
    // x sub-bins (kernels.hpp XSubBin; untyped boxes): nsub = 4 when the sort orders a cell's atoms by quarter, sub_k = K,
    // fstart = first slot of every (cell, sub-bin) block, bsub = per brick and tile cell the three inner boundaries of the
    // cell's sub-bins (10 bits each, relative to the cell's first atom), written by k_brick_tables for the build kernel
    int nsub, sub_k;
    const int *fstart;
    int *bsub;
    // near/far rows with the far class skipped outright (round 5 experiment, EMDEE_FAR_SKIP=1 on top of EMDEE_BUILD_NEARFAR=1):
    // cnt[p] = total | near << 8, and while NO atom has moved delta / 2 since the build (*far_word == 0: raised like the
    // rebuild trigger, by the launch that produced the positions, read by the next one) a row ends at its near entries -- an
    // entry beyond r_c + delta at the build cannot be inside r_c before two atoms have moved delta / 2 each: the same sums
    // cell-relative records (kernels.hpp RelGrid; fp32 integrators): a record is relative to the origin of its cell, and a tile
    // coordinate is record + (tile cell - brick origin) cell widths -- no image shift, no box-sized number anywhere
    int rel;
    double rcw[3], rlo[3];
    int far_skip;
    int *far_word;
    real thr2_near;
    const real *user_pos;      // ... read from the CALLER's array (3 x N, caller order): the engine's records hold positions wrapped
                               // into the box, and x - L rounded to fp32 is not the number the reference divides by L
};

// ---- cell-relative records: tile coordinates (kernels.hpp RelGrid) ---------------------------------
// the integer (as a float: < 2^24) that takes round(record x 2^19) of an atom of the tile cell at offset t (-1 .. B) from the
// brick's first own cell b, along dimension d, to its brick-relative tile coordinate in grid points
// (M, lo, cw of ONE dimension are passed as values: indexing the kernel-argument struct with a run-time dimension makes the compiler
// keep a copy of the whole struct in scratch memory -- the fp32 two-species kernel ran at 5.0 instead of 2.2 ms for that, found in round 5)
__device__ __forceinline__ float rel_cell_const(int M, double lo, double cw, int b, int t) {
    int cg = b + t, img = 0;                                 // the cell of the box this tile cell is an image of
    if (cg < 0) { cg += M; img = -1; } else if (cg >= M) { cg -= M; img = 1; }
    auto oq = [&](int c) { return rint((lo + (double)c * cw) * 524288.0); };
    return (float)(oq(cg) + (double)img * oq(M) - (double)img * oq(0) - oq(b));
}
// ... for the TX + TY + TZ tile-cell offsets of a brick, into LDS (56 bytes): call with every thread, then a barrier
template <typename real, class Shape>
__device__ __forceinline__ void rel_fill_consts(const BrickArgs<real> &a, int bxi, int byi, int bzi, float *relc) {
    constexpr int TX = Shape::TX, TY = Shape::TY, TZ = Shape::TZ;
    const int t = threadIdx.x;
    if (t < TX) relc[t] = rel_cell_const(a.g.M[0], a.rlo[0], a.rcw[0], bxi * Shape::BX, t - 1);
    else if (t < TX + TY) relc[t] = rel_cell_const(a.g.M[1], a.rlo[1], a.rcw[1], byi * Shape::BY, t - TX - 1);
    else if (t < TX + TY + TZ) relc[t] = rel_cell_const(a.g.M[2], a.rlo[2], a.rcw[2], bzi * Shape::BZ, t - TX - TY - 1);
}
__device__ __forceinline__ float rel_tile(float v, float c) { return (rintf(v * REL_FX) + c) * REL_IFX; }
__device__ __forceinline__ double rel_tile(double v, float) { return v; }   // (fp64 states are never cell-relative)

// ---- LDS tables shared by the build and force kernels ------------------------------------------
template <class Shape, int THREADS>
struct BrickTables {
    static constexpr int NTC = Shape::NTC, NOC = Shape::NOC, NWAVES = THREADS / WAVE;
    int *off;      // [NTC+1] tile-local first slot of each tile cell
    int *gbeg;     // [NTC]   global (cell-order) first slot of each tile cell
    int *shift;    // [NTC]   periodic image: 2 bits per dimension (0:-1, 1:0, 2:+1)
    int *own;      // [NOC+1] prefix of own-cell populations
    int *wtot;     // [NWAVES]
    int2 *oinfo;   // [own_cap] per own atom {cell-order slot p, (row length or flag) << 16 | tile slot}
    __host__ __device__ static constexpr size_t fixed_ints() { return (NTC + 4) + NTC + NTC + (NOC + 4) + ((NWAVES + 1) & ~1); }
    __host__ __device__ static size_t bytes(int own_cap) { return ((fixed_ints() + 2 * (size_t)own_cap) * 4 + 15) & ~(size_t)15; }
    // off | gbeg | shift | own are contiguous: that image (+ tile_n, n_own) is what k_brick_tables stores per brick
    __host__ __device__ static constexpr int image_ints() { return (NTC + 4) + NTC + NTC + (NOC + 4); }
    __host__ __device__ static constexpr int row_ints() { return (image_ints() + 2 + 3) & ~3; }
    __device__ __forceinline__ void carve(unsigned char *base) {
        off = reinterpret_cast<int *>(base);
        gbeg = off + (NTC + 4);
        shift = gbeg + NTC;
        own = shift + NTC;
        wtot = own + (NOC + 4);
        oinfo = reinterpret_cast<int2 *>(wtot + ((NWAVES + 1) & ~1));
    }
};

// Single-species fp64 force kernels keep only the three coordinate planes in LDS (24 B per record instead of the
// 32-byte HBM record): with a fixed plane pitch the three reads of a neighbour share one address register and differ
// in the instruction's immediate offset, and three workgroups fit a CU where the 32-byte tile allows two.
constexpr int SOA_SLOTS = 2048;                      // records a coordinate-plane tile can hold
// Plane pitch in records.  fp64: one record more than a power of two, so that the x and y reads of a neighbour cannot be
// merged into one ds_read2st64_b64 (8 LDS-array cycles per wavefront, MI355X_MICROARCH.md LDS table) and stay two
// ds_read_b64 (2 cycles each) off one address register with immediate offsets.
#ifndef EMDEE_SOA_PAD
#define EMDEE_SOA_PAD 1
#endif
template <typename real>
constexpr int soa_pitch() { return SOA_SLOTS + ((sizeof(real) == 8 && EMDEE_SOA_PAD) ? 1 : 0); }
template <typename real, class Shape, int THREADS>
static inline size_t brick_force_lds_bytes_soa(int own_cap) {
    return (((size_t)3 * soa_pitch<real>() * sizeof(real) + 15) & ~(size_t)15) + BrickTables<Shape, THREADS>::bytes(own_cap);
}

// bytes of dynamic LDS: force tile = HBM records (+ te plane for fp32); build tile = float4
template <typename real, class Shape, int THREADS>
static inline size_t brick_force_lds_bytes(int tile_cap, int own_cap) {
    size_t tile_bytes = (size_t)tile_cap * sizeof(Rec<real>);
    size_t te_bytes = sizeof(real) == 4 ? (((size_t)tile_cap * 4 + 15) & ~(size_t)15) : 0;
    return tile_bytes + te_bytes + BrickTables<Shape, THREADS>::bytes(own_cap);
}
template <class Shape, int THREADS>
static inline size_t brick_build_lds_bytes(int tile_cap, int own_cap, int stride, int G, int nsub = 1) {
    // + one row buffer (stride uint16) per G-lane group
    // + the candidate rows of every own cell ({first slot, span} of its 9 tile rows, and one all-zero set for ghosts)
    // (one packed word per row; nsub > 1: one set of rows per own cell AND sub-bin)
    return (size_t)tile_cap * 16 + BrickTables<Shape, THREADS>::bytes(own_cap) + (size_t)(THREADS / G) * stride * 2 +
           (((size_t)(Shape::NOC * nsub + 1) * 9 * 4 + 15) & ~(size_t)15);
}
// atoms of a tile cell in sub-bins below s, from its packed boundaries (s <= 0: none, s >= 4: the whole cell)
__device__ __forceinline__ int sub_below(int packed, int s, int population) {
    return s <= 0 ? 0 : (s >= 4 ? population : ((packed >> (10 * (s - 1))) & 1023));
}

// position of the e-th neighbour inside a row: blocks of 8 G entries, lane-major inside a block,
// so lane l of the group finds entries l, l+G, ..., l+7G of the block in 8 consecutive uint16
template <int G>
__host__ __device__ __forceinline__ unsigned row_position(unsigned e) {
    constexpr unsigned BLK = EPL * G;   // all powers of two: three masks/shifts
    return (e & ~(BLK - 1u)) | ((e & (G - 1u)) * EPL) | ((e / G) & (EPL - 1u));
}

// max over the groups of a wavefront of a group-uniform, non-negative value, as a scalar: DPP row shifts and row
// broadcasts (no LDS round trips -- the pair loop of a round cannot start before it knows its trip count)
template <int G>
__device__ __forceinline__ int wave_group_max(int v) {
    if (G <= 1) v = max(v, __builtin_amdgcn_update_dpp(0, v, DPP_ROW_SHR1, 0xf, 0xf, true));
    if (G <= 2) v = max(v, __builtin_amdgcn_update_dpp(0, v, DPP_ROW_SHR2, 0xf, 0xf, true));
    if (G <= 4) v = max(v, __builtin_amdgcn_update_dpp(0, v, DPP_ROW_SHR4, 0xf, 0xf, true));
    if (G <= 8) v = max(v, __builtin_amdgcn_update_dpp(0, v, DPP_ROW_SHR8, 0xf, 0xf, true));
    if (G <= 16) v = max(v, __builtin_amdgcn_update_dpp(0, v, DPP_ROW_BCAST15, 0xa, 0xf, true));
    if (G <= 32) v = max(v, __builtin_amdgcn_update_dpp(0, v, DPP_ROW_BCAST31, 0xc, 0xf, true));
    return __builtin_amdgcn_readlane(v, WAVE - 1);
}

__device__ __forceinline__ int pick16(const uint4 &q, int t) {   // t is a compile-time constant after unrolling
    const unsigned w = (t < 2) ? q.x : (t < 4) ? q.y : (t < 6) ? q.z : q.w;
    return (int)((t & 1) ? (w >> 16) : (w & 0xffffu));
}

// Which brick does this workgroup own?  (phase 0: every brick, XCD-contiguous; 1: the interior sub-box; 2: the boundary
// shell.)  False if the block is past the end of its enumeration.
template <typename real>
__device__ __forceinline__ bool brick_of_block(const BrickArgs<real> &a, int &bxi, int &byi, int &bzi) {
    if (a.phase == 1) {
        // interior bricks only, enumerated densely: the grid is ib_per_xcd * 8 blocks
        const int li = (blockIdx.x % NXCD) * a.bg.ib_per_xcd + blockIdx.x / NXCD;
        if (li >= a.bg.ib_count) return false;
        bxi = a.bg.ib_lo[0] + li % a.bg.ib_n[0];
        byi = a.bg.ib_lo[1] + (li / a.bg.ib_n[0]) % a.bg.ib_n[1];
        bzi = a.bg.ib_lo[2] + li / (a.bg.ib_n[0] * a.bg.ib_n[1]);
    } else if (a.phase == 2) {
        // boundary bricks only (their tile reaches the outermost cell layer of an open dimension, where
        // the ghosts of a decomposed run live), enumerated densely
        int li = (blockIdx.x % NXCD) * a.bg.bb_per_xcd + blockIdx.x / NXCD;
        if (li >= a.bg.bb_count) return false;
        const int nx = a.bg.nb[0], ny = a.bg.nb[1];
        if (li < a.bg.bb_z) {                                   // whole xy planes outside the interior z range
            bxi = li % nx; byi = (li / nx) % ny;
            bzi = outside_interval(li / (nx * ny), a.bg.ib_lo[2], a.bg.ib_n[2]);
        } else if (li < a.bg.bb_z + a.bg.bb_y) {                // interior z, y outside
            li -= a.bg.bb_z;
            const int oy = ny - a.bg.ib_n[1];
            bxi = li % nx; byi = outside_interval((li / nx) % oy, a.bg.ib_lo[1], a.bg.ib_n[1]);
            bzi = a.bg.ib_lo[2] + li / (nx * oy);
        } else {                                                // interior z and y, x outside
            li -= a.bg.bb_z + a.bg.bb_y;
            const int ox = nx - a.bg.ib_n[0];
            bxi = outside_interval(li % ox, a.bg.ib_lo[0], a.bg.ib_n[0]);
            byi = a.bg.ib_lo[1] + (li / ox) % a.bg.ib_n[1];
            bzi = a.bg.ib_lo[2] + li / (ox * a.bg.ib_n[1]);
        }
    } else {
        const int lb = (blockIdx.x % NXCD) * a.bg.per_xcd + blockIdx.x / NXCD;   // XCD-contiguous brick order
        if (lb >= a.bg.nbricks) return false;
        bxi = lb % a.bg.nb[0]; byi = (lb / a.bg.nb[0]) % a.bg.nb[1]; bzi = lb / (a.bg.nb[0] * a.bg.nb[1]);
    }
    return true;
}

// Fills the tile-cell and own-cell tables of this block's brick.  Returns false when the block
// has nothing to do.  Contains block barriers: every thread of the block must call it.
template <typename real, class Shape, int THREADS, bool COMPUTE = false>
__device__ __forceinline__ bool brick_setup(const BrickArgs<real> &a, const BrickTables<Shape, THREADS> &T, int &bxi,
                                            int &byi, int &bzi, int &tile_n, int &n_own) {
    constexpr int BX = Shape::BX, BY = Shape::BY, BZ = Shape::BZ, TX = Shape::TX, TY = Shape::TY, NTC = Shape::NTC,
                  NOC = Shape::NOC;
    const int tid = threadIdx.x, lane = tid & (WAVE - 1), wv = tid / WAVE;
    const int Mx = a.g.M[0], My = a.g.M[1], Mz = a.g.M[2];
    if (!brick_of_block(a, bxi, byi, bzi)) return false;
    if (!COMPUTE && a.btab != nullptr) {
        // the tables of this brick were computed when the list was built (the cell populations are frozen until the
        // next rebuild): copy the image instead of redoing the scans and the serial own-cell prefix in every launch
        constexpr int ROW = BrickTables<Shape, THREADS>::row_ints(), IMG = BrickTables<Shape, THREADS>::image_ints();
        const int *row = a.btab + (size_t)(bxi + a.bg.nb[0] * (byi + a.bg.nb[1] * bzi)) * ROW;
        for (int i = tid; i < ROW / 4; i += THREADS)
            reinterpret_cast<uint4 *>(T.off)[i] = reinterpret_cast<const uint4 *>(row)[i];
        __syncthreads();
        tile_n = T.off[IMG];
        n_own = T.off[IMG + 1];
        // (capacities may come from the plan of the previous rebuild: a brick that outgrew them is skipped and reported,
        // the host plans afresh and builds again before any force launch)
        if (tile_n > a.tile_cap || n_own > a.own_cap) {
            if (tid == 0) atomicMax(&a.flags[2], max(tile_n, n_own));
            return false;
        }
        return n_own > 0;
    }
    int my_cnt = 0;
    if (tid < NTC) {
        const int tx = tid % TX, ty = (tid / TX) % TY, tz = tid / (TX * TY);
        int gx = bxi * BX - 1 + tx, gy = byi * BY - 1 + ty, gz = bzi * BZ - 1 + tz;
        // own range may be clipped on the high side of a partial brick; halo = own range +- 1
        const int ox1 = min(bxi * BX + BX, Mx), oy1 = min(byi * BY + BY, My), oz1 = min(bzi * BZ + BZ, Mz);
        bool valid = gx <= ox1 && gy <= oy1 && gz <= oz1;
        int sh = 1 | (1 << 2) | (1 << 4);
        auto wrap = [&](int &c, int M, int per, int bit) {
            if (c < 0) {
                if (per) { c += M; sh = (sh & ~(3 << bit)) | (0 << bit); } else valid = false;
            } else if (c >= M) {
                if (per) { c -= M; sh = (sh & ~(3 << bit)) | (2 << bit); } else valid = false;
            }
        };
        wrap(gx, Mx, a.g.per[0], 0);
        wrap(gy, My, a.g.per[1], 2);
        wrap(gz, Mz, a.g.per[2], 4);
        int gb = 0;
        if (valid) {
            const int c = gx + Mx * (gy + My * gz);
            gb = a.start[c];
            my_cnt = a.start[c + 1] - gb;
        }
        // Decomposed runs: an own cell that holds nothing but ghosts (the outermost cell layers of a cut dimension, all
        // but a sliver of them) takes no part in the own-atom loops of the build and force kernels -- ghosts own no row,
        // receive no force and are not integrated, but as own atoms they occupied one lane group each: 16 % of the atoms
        // of a rank of the 8-GPU 10^7-atom run.  Marked with bit 8 of the cell's shift word (part of the stored image).
        if (COMPUTE && a.any_ghosts && valid && my_cnt > 0 && tx >= 1 && tx <= BX && ty >= 1 && ty <= BY && tz >= 1 && tz <= BZ) {
            int owned = 0;
            for (int k = 0; k < my_cnt; k++) owned |= (a.perm[gb + k] < a.n_owned) ? 1 : 0;
            if (!owned) sh |= 256;
        }
        T.gbeg[tid] = gb;
        T.shift[tid] = sh;
        if (COMPUTE && a.bsub != nullptr) {
            int packed = 0;
            if (valid && a.nsub == 4) {
                const int *fs = a.fstart + 4 * (size_t)(gx + Mx * (gy + My * gz));
                packed = (fs[1] - gb) | ((fs[2] - gb) << 10) | ((fs[3] - gb) << 20);
            }
            a.bsub[(size_t)(bxi + a.bg.nb[0] * (byi + a.bg.nb[1] * bzi)) * NTC + tid] = packed;
        }
    }
    {   // exclusive scan of my_cnt over the first NTC threads (NTC <= THREADS)
        int inc = my_cnt;
#pragma unroll
        for (int off = 1; off < WAVE; off <<= 1) {
            int t = __shfl_up(inc, off);
            if (lane >= off) inc += t;
        }
        if (lane == WAVE - 1) T.wtot[wv] = inc;
        __syncthreads();
        int woff = 0;
        for (int w = 0; w < wv; w++) woff += T.wtot[w];
        // tile slots start at 1: slot 0 is the SENTINEL record (parked far outside the box by the force kernel),
        // so an index word that was never loaded (all zero bits) and the padding of a row mean the same thing
        if (tid < NTC) T.off[tid] = 1 + woff + inc - my_cnt;
        if (tid == NTC - 1) T.off[NTC] = 1 + woff + inc;
    }
    __syncthreads();
    tile_n = T.off[NTC];
    if (tid == 0) {
        int acc = 0;
        for (int oc = 0; oc < NOC; oc++) {
            const int ox = oc % BX, oy = (oc / BX) % BY, oz = oc / (BX * BY);
            const int tc = (ox + 1) + TX * ((oy + 1) + TY * (oz + 1));
            T.own[oc] = acc;
            // cells past the box edge of a partial brick hold halo images, not atoms of this brick
            const bool mine = (bxi * BX + ox < Mx) && (byi * BY + oy < My) && (bzi * BZ + oz < Mz) && !(T.shift[tc] & 256);
            acc += mine ? (T.off[tc + 1] - T.off[tc]) : 0;
        }
        T.own[NOC] = acc;
    }
    __syncthreads();
    n_own = T.own[NOC];
    if (tile_n > a.tile_cap || n_own > a.own_cap) {   // cannot happen: the host sized both from k_brick_tile_max
        if (tid == 0) atomicMax(&a.flags[2], max(tile_n, n_own));
        return false;
    }
    return n_own > 0;
}

// own atom o -> own cell, tile slot, cell-order slot
template <class Shape, int THREADS>
__device__ __forceinline__ int brick_locate(const BrickTables<Shape, THREADS> &T, int o, int &ti, int &p) {
    int oc = 0;
#pragma unroll
    for (int q = 1; q < Shape::NOC; q++) oc += (T.own[q] <= o) ? 1 : 0;
    const int ox = oc % Shape::BX, oy = (oc / Shape::BX) % Shape::BY, oz = oc / (Shape::BX * Shape::BY);
    const int tc = (ox + 1) + Shape::TX * ((oy + 1) + Shape::TY * (oz + 1));
    const int kk = o - T.own[oc];
    ti = T.off[tc] + kk;
    p = T.gbeg[tc] + kk;
    return oc;
}

// Calls f(slot, tile cell) once for every tile slot.  Half-waves take whole tile rows (TX cells,
// contiguous in the tile and, away from a periodic wrap, in HBM): consecutive lanes touch consecutive
// records, and a slot's cell is found with TX-1 loop-invariant compares instead of a dependent
// binary search over all tile cells.
constexpr int STAGE_LANES = 32;
template <class Shape, int THREADS, class F>
__device__ __forceinline__ void brick_for_each_slot(const BrickTables<Shape, THREADS> &T, F &&f) {
    constexpr int NW = THREADS / STAGE_LANES, NROWS = Shape::TY * Shape::TZ, TX = Shape::TX;
    const int worker = threadIdx.x / STAGE_LANES, l = threadIdx.x % STAGE_LANES;
    for (int row = worker; row < NROWS; row += NW) {
        const int c0 = row * TX;
        const int ty = row % Shape::TY, tz = row / Shape::TY;      // (per row, not per slot: the cell-relative staging indexes by them)
        int edge[TX + 1];
#pragma unroll
        for (int c = 0; c <= TX; c++) edge[c] = T.off[c0 + c];
        for (int s = edge[0] + l; s < edge[TX]; s += STAGE_LANES) {
            int tx = 0;
#pragma unroll
            for (int c = 1; c < TX; c++) tx += (edge[c] <= s) ? 1 : 0;
            f(s, c0 + tx, tx, ty, tz);
        }
    }
}

// Once per rebuild: the tables of every brick, stored as the LDS image the other kernels copy in (brick_setup).
template <typename real, class Shape, int THREADS>
__global__ __launch_bounds__(THREADS) void k_brick_tables(BrickArgs<real> a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char s_dyn[];
    BrickTables<Shape, THREADS> T;
    T.carve(s_dyn);
    const int lb = (blockIdx.x % NXCD) * a.bg.per_xcd + blockIdx.x / NXCD;
    if (lb >= a.bg.nbricks) return;
    int bxi = 0, byi = 0, bzi = 0, tile_n = 0, n_own = 0;
    brick_setup<real, Shape, THREADS, true>(a, T, bxi, byi, bzi, tile_n, n_own);   // false also for empty bricks: stored too
    constexpr int ROW = BrickTables<Shape, THREADS>::row_ints(), IMG = BrickTables<Shape, THREADS>::image_ints();
    int *row = a.btab + (size_t)(bxi + a.bg.nb[0] * (byi + a.bg.nb[1] * bzi)) * ROW;
    for (int i = threadIdx.x; i < IMG; i += THREADS) row[i] = T.off[i];
    if (threadIdx.x == 0) { row[IMG] = tile_n; row[IMG + 1] = n_own; }
    // The population maxima a kept plan is checked against (NbSystem::plan_holds), as k_brick_tile_max reports them:
    // flags[6] largest tile, [7] most own atoms, [8] most atoms in three consecutive cells of a tile row.
    if (a.stats != nullptr) {
        int span3 = 0;
        if (threadIdx.x < Shape::NTC && threadIdx.x % Shape::TX + 3 <= Shape::TX) span3 = T.off[threadIdx.x + 3] - T.off[threadIdx.x];
#pragma unroll
        for (int off = 1; off < WAVE; off <<= 1) span3 = max(span3, __shfl_xor(span3, off));
        // (same-address atomics serialise in L2 at ~10 ns each and there is one workgroup per brick: look first (a device-scope load: the per-CU cache would
        // keep showing the zero it saw first), and only the few bricks that raise a maximum pay for an atomic -- without the look this kernel took 1.6 ms instead of 0.04)
        int *maxima = reinterpret_cast<int *>(a.stats);
        if ((threadIdx.x & (WAVE - 1)) == 0 && span3 > __hip_atomic_load(&maxima[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(&maxima[2], span3);
        if (threadIdx.x == 0) {
            if (tile_n - 1 > __hip_atomic_load(&maxima[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(&maxima[0], tile_n - 1);
            if (n_own > __hip_atomic_load(&maxima[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(&maxima[1], n_own);
        }
    }
}

constexpr int OWN_REGS = 2;   // own atoms per thread whose table entry is fetched before the tile is staged

// ------------------------------------------------------------------------------------ build
// ALG 1: candidates are dealt round-robin to the lanes of a group and every pass of the test loop
// compacts its hits with a ballot (count, prefix, scattered 2-byte LDS store: more VALU work than the
// distance test itself).  ALG 2 (default when no 3-cell tile row holds more than BUILD2_MAX_SPAN atoms):
// each lane tests a CONTIGUOUS chunk of every tile row and only records one bit per candidate; the hits
// are compacted once per atom, from the bit fields, after a prefix sum over the lanes of the group.  Rows
// come out ordered lane by lane, i.e. still along the tile rows, so neighbouring entries keep pointing at
// neighbouring tile slots.
constexpr int BUILD2_FIELD = 16;                      // bits per tile row in a lane's bit field

// G = lanes that share one atom HERE; GL = lanes per atom of the force kernels, which fixes the lane-major row
// layout (row_position<GL>) -- the two need not agree.
template <typename real, class Shape, int THREADS, int G, int ALG = 1, int GL = G>
__global__ __launch_bounds__(THREADS, (THREADS <= 768 ? 6 : 4)) void k_brick_build(BrickArgs<real> a) {   // <= 80 VGPRs: three 512-thread (or two 768-thread) workgroups per CU
    constexpr int BX = Shape::BX, BY = Shape::BY, TX = Shape::TX, TY = Shape::TY;
    constexpr int NGROUPS = (THREADS / WAVE) * (WAVE / G);
    extern __shared__ __attribute__((aligned(16))) unsigned char s_dyn[];
    float4 *tile = reinterpret_cast<float4 *>(s_dyn);   // {x, y, z relative to the brick origin, cell-order slot}
    BrickTables<Shape, THREADS> T;
    T.carve(s_dyn + (size_t)a.tile_cap * 16);
    int bxi, byi, bzi, tile_n, n_own;
    if (!brick_setup<real, Shape, THREADS>(a, T, bxi, byi, bzi, tile_n, n_own)) return;
    const int tid = threadIdx.x, lane = tid & (WAVE - 1);
    // x sub-bins (the two-phase builds that take their candidate rows from the row table): rows per own cell AND sub-bin
    constexpr bool SUBOK = ALG == 3 || ALG == 5 || ALG == 13 || ALG == 15 || ALG == 23;
    const int NSUB = (SUBOK && a.nsub == 4 && a.bsub != nullptr) ? 4 : 1;
    // (one word per row: first slot | slots << 16 -- a tile holds < 2^16 slots; the build's LDS decides whether a CU takes
    // three workgroups.  The sub-bin boundaries of the tile cells are only needed until the row table is written: they
    // borrow the row buffers, which nobody touches before the barrier behind that.)
    unsigned *rtab = reinterpret_cast<unsigned *>(s_dyn + (size_t)a.tile_cap * 16 + BrickTables<Shape, THREADS>::bytes(a.own_cap) +
                                                  (size_t)NGROUPS * a.stride * 2);
    int *s_sub = reinterpret_cast<int *>(s_dyn + (size_t)a.tile_cap * 16 + BrickTables<Shape, THREADS>::bytes(a.own_cap));
    auto rt_get = [](const unsigned *rt, int r) { const unsigned v = rt[r]; return make_int2((int)(v & 0xffffu), (int)(v >> 16)); };
    if (NSUB > 1) {
        const int *bs = a.bsub + (size_t)(bxi + a.bg.nb[0] * (byi + a.bg.nb[1] * bzi)) * Shape::NTC;
        for (int i = tid; i < Shape::NTC; i += THREADS) s_sub[i] = bs[i];
        __syncthreads();
    }
    // sub-bin of the own atom at tile slot ti of own cell oc
    auto own_sub = [&](int oc, int ti) {
        if (NSUB == 1) return 0;
        const int tc = (oc % BX + 1) + TX * (((oc / BX) % BY + 1) + TY * (oc / (BX * BY) + 1));
        const int kk = ti - T.off[tc], pk = s_sub[tc];
        return (kk >= (pk & 1023) ? 1 : 0) + (kk >= ((pk >> 10) & 1023) ? 1 : 0) + (kk >= ((pk >> 20) & 1023) ? 1 : 0);
    };

    // brick origin: fp64 boxes are re-based here so that fp32 coordinates stay small
    real org[3] = {0, 0, 0};
    if (sizeof(real) == 8) {
        org[0] = a.g.lo[0] + (real)(bxi * BX) * (a.g.len[0] / (real)a.g.M[0]);
        org[1] = a.g.lo[1] + (real)(byi * BY) * (a.g.len[1] / (real)a.g.M[1]);
        org[2] = a.g.lo[2] + (real)(bzi * Shape::BZ) * (a.g.len[2] / (real)a.g.M[2]);
    }
    __shared__ float s_relc[sizeof(real) == 4 ? Shape::TX + Shape::TY + Shape::TZ : 1];   // cell-relative records: rel_cell_const per tile-cell offset
    if (sizeof(real) == 4 && a.rel) {
        rel_fill_consts<real, Shape>(a, bxi, byi, bzi, s_relc);
        __syncthreads();
    }
    // own-atom table entries first (their global loads fly while the tile is being staged)
    int own_p[OWN_REGS], own_ti[OWN_REGS], own_key[OWN_REGS], own_oc[OWN_REGS];
#pragma unroll
    for (int k = 0; k < OWN_REGS; k++) {
        const int o = tid + k * THREADS;
        own_p[k] = own_ti[k] = 0; own_key[k] = 0; own_oc[k] = 0;
        if (o < n_own) {
            own_oc[k] = brick_locate(T, o, own_ti[k], own_p[k]);
            own_key[k] = a.perm[own_p[k]];
        }
    }
    brick_for_each_slot(T, [&](int s, int tc, int tx, int ty, int tz) {
        const int gp = T.gbeg[tc] + (s - T.off[tc]);
        const int sh = T.shift[tc];
        const Rec<real> r = a.rec[gp];
        float4 q;
        if constexpr (ALG == 23) {   // (scaled in the box's own precision, one rounding to fp32 as before)
            const real ks = (real)a.nf_scale;
            q.x = (float)(((r.x + (real)((sh & 3) - 1) * a.g.len[0]) - org[0]) * ks);
            q.y = (float)(((r.y + (real)(((sh >> 2) & 3) - 1) * a.g.len[1]) - org[1]) * ks);
            q.z = (float)(((r.z + (real)(((sh >> 4) & 3) - 1) * a.g.len[2]) - org[2]) * ks);
        } else if (sizeof(real) == 4 && a.rel) {              // cell-relative records: the tile coordinates the force kernels see, to the bit
            q.x = (float)rel_tile(r.x, s_relc[tx]);
            q.y = (float)rel_tile(r.y, s_relc[TX + ty]);
            q.z = (float)rel_tile(r.z, s_relc[TX + TY + tz]);
        } else {
            q.x = (float)((r.x + (real)((sh & 3) - 1) * a.g.len[0]) - org[0]);
            q.y = (float)((r.y + (real)(((sh >> 2) & 3) - 1) * a.g.len[1]) - org[1]);
            q.z = (float)((r.z + (real)(((sh >> 4) & 3) - 1) * a.g.len[2]) - org[2]);
        }
        q.w = __int_as_float(gp);
        if (EMDEE_BOUND(BS_BUILD_TILE, s, a.tile_cap)) tile[s] = q;
    });
#pragma unroll
    for (int k = 0; k < OWN_REGS; k++) {
        const int o = tid + k * THREADS;
        if (o < n_own && EMDEE_BOUND(BS_BUILD_OWN, o, a.own_cap))
            T.oinfo[o] = make_int2(own_p[k], (own_oc[k] << 20) | (own_sub(own_oc[k], own_ti[k]) << 17) | ((own_key[k] < a.n_owned ? 1 : 0) << 16) | own_ti[k]);
    }
    for (int o = tid + OWN_REGS * THREADS; o < n_own; o += THREADS) {   // very dense bricks only
        int ti, p;
        const int oc = brick_locate(T, o, ti, p);
        if (EMDEE_BOUND(BS_BUILD_OWN, o, a.own_cap)) T.oinfo[o] = make_int2(p, (oc << 20) | (own_sub(oc, ti) << 17) | ((a.perm[p] < a.n_owned ? 1 : 0) << 16) | ti);
    }
    // candidate rows per own cell (and sub-bin): the 9 tile rows (dy, dz) of 3 cells around it, as {first tile slot, slots};
    // the last entry is empty (atoms that own no row).  One table per brick instead of index arithmetic and two reads per
    // atom and row.  With x sub-bins (cells sorted by quarter along x) an atom of sub-bin s takes the quarters >= s + K of
    // the left cell, the whole middle cell and the quarters <= s - K of the right cell: still one run of tile slots.
    for (int i = tid; i < (Shape::NOC * NSUB + 1) * 9; i += THREADS) {
        const int ocs = i / 9, r = i % 9;
        unsigned v = 0;
        if (ocs < Shape::NOC * NSUB) {
            const int oc = ocs / NSUB, sb = ocs - oc * NSUB;
            const int ox = oc % BX, oy = (oc / BX) % BY, oz = oc / (BX * BY);
            const int tcr = ox + TX * ((oy + r % 3) + TY * (oz + r / 3));       // cell x-1 of tile row (dy, dz) = (r%3-1, r/3-1)
            int first = T.off[tcr], last = T.off[tcr + 3];
            if (NSUB > 1) {
                first += sub_below(s_sub[tcr], sb + a.sub_k, T.off[tcr + 1] - T.off[tcr]);
                last = T.off[tcr + 2] + sub_below(s_sub[tcr + 2], sb - a.sub_k + 1, T.off[tcr + 3] - T.off[tcr + 2]);
            }
            v = (unsigned)first | ((unsigned)(last - first) << 16);
        }
        rtab[i] = v;
    }
    __syncthreads();

    const int gl = lane & (G - 1);
    const int gid = (tid / WAVE) * (WAVE / G) + lane / G;
    const float rl2 = (float)a.rlist2;
    const float lo2 = rl2 - a.margin, hi2 = rl2 + a.margin;   // margin == 0 for fp32 boxes: the fp32 test is exact
    // Each group assembles its row in LDS (pre-filled with the SENTINEL slot 0, a record the force
    // kernel parks far outside the box, so that it can walk whole blocks with no per-lane bound test) and
    // writes it out as 16-byte, fully coalesced stores instead of scattered 2-byte ones.
    unsigned short *rowbuf = reinterpret_cast<unsigned short *>(s_dyn + (size_t)a.tile_cap * 16 +
                                                                BrickTables<Shape, THREADS>::bytes(a.own_cap)) +
                             (size_t)gid * a.stride;
    // ballot bits of my group, in 32-bit arithmetic (a group never straddles the two halves of the mask)
    static_assert(G <= 32, "build kernel: groups of at most 32 lanes");
    const unsigned gshift = (unsigned)(lane & 31 & ~(G - 1));
    const unsigned gmask = G == 32 ? 0xffffffffu : ((1u << G) - 1u);
    const unsigned ltmask = (1u << gl) - 1u;
    const bool upper = lane >= 32;
    const unsigned ustride = (unsigned)a.stride;
    const uint4 fill = make_uint4(0, 0, 0, 0);               // sentinel slot 0 (see brick_setup)
    // fp32 test of candidate slot c against atom (p, qi); pairs inside the rounding band are decided with
    // the exact fp64 records (tcr = tile cell x-1 of the candidate's tile row)
    auto in_range_q = [&](const float4 &qi, const float4 &qj, int p, int c, int tcr) -> bool {
        const float dx = qi.x - qj.x, dy2 = qi.y - qj.y, dz2 = qi.z - qj.z;
        const float d2 = dx * dx + dy2 * dy2 + dz2 * dz2;
        bool pass = d2 < lo2;
        if (sizeof(real) == 8 && __builtin_expect(!pass && d2 <= hi2, 0)) {
            const int tc = tcr + (c >= T.off[tcr + 1] ? 1 : 0) + (c >= T.off[tcr + 2] ? 1 : 0);
            const int sh = T.shift[tc];
            const Rec<real> ri = a.rec[p], rj = a.rec[__float_as_int(qj.w)];
            const real ex = ri.x - (rj.x + (real)((sh & 3) - 1) * a.g.len[0]);
            const real ey = ri.y - (rj.y + (real)(((sh >> 2) & 3) - 1) * a.g.len[1]);
            const real ez = ri.z - (rj.z + (real)(((sh >> 4) & 3) - 1) * a.g.len[2]);
            pass = ex * ex + ey * ey + ez * ez < a.rlist2;
        }
        return pass;
    };
    auto in_range = [&](const float4 &qi, int p, int c, int tcr) -> bool { return in_range_q(qi, tile[c], p, c, tcr); };
    if constexpr (ALG == 23) {
        // ---- ALG 13 with NEAR entries first --------------------------------------------------------------------------
        // The force kernels run their 31-instruction pair body whenever ANY of the 64 lanes of a wavefront is inside the
        // cutoff, and 29 % of a row are skin entries (r_c <= d < r_list when the list is built).  Rows are therefore
        // written near entries first (d < r_near = r_c + delta), far entries behind them: the lanes of a wavefront walk
        // their rows in step, so its last pair steps see far entries in every lane -- pairs that only come inside the
        // cutoff if both atoms use up most of the skin -- fail the cutoff test wave-wide and skip the body.  The force
        // kernels need no change (the cutoff test decides, as ever); only the ORDER of a row differs.
        // The class of a candidate costs nothing: with the tile scaled by k, k^2 (r_near^2 - r_list^2) = -2.0, the top two
        // bits of the float t = k^2 (d^2 - r_list^2) are {t < 0 : listed, |t| >= 2 : near}, and ONE v_alignbit_b32 shifts
        // both into the lane's digit string (the two-instruction compare + add-with-carry of ALG 13 did one bit).
        static_assert(G == 8 && sizeof(unsigned) == 4, "near/far build: 8 lanes per atom");
        constexpr int NROWS = 9, LOG2G = 3;
        constexpr bool BAND = sizeof(real) == 8;
        float nrl2 = -rl2 * a.nf_scale2, margin_v = a.margin * a.nf_scale2;
        asm volatile("" : "+v"(nrl2), "+v"(margin_v));
        const int kshift = a.idx_shift + LOG2G;                  // digit j of a row is tile slot cb + (trips - 1 - j) G
        for (int ob = 0; ob < n_own; ob += NGROUPS) {            // wave-uniform trip count
            const int o = ob + gid;
            const bool have = o < n_own;
            const int2 info = have ? T.oinfo[o] : make_int2(0, 0);
            const int ti = info.y & 0xffff, p = info.x, oc = info.y >> 20;
            const bool act = have && ((info.y >> 16) & 1) != 0;   // ghosts own no row
            const float4 qi = tile[ti];
            unsigned short *row = a.nbr + (size_t)p * a.stride;
            for (int c = gl * EPL; c < a.stride; c += G * EPL) *reinterpret_cast<uint4 *>(rowbuf + c) = fill;
            const unsigned *rt = rtab + (act ? oc * NSUB + ((info.y >> 17) & 3) : Shape::NOC * NSUB) * 9;   // (x sub-bins: as ALG 13)
            int trips_of[NROWS];
            {   // wave-uniform trip counts of the nine rows from one reduction (as ALG 13)
                int cv = (int)((unsigned)(rt_get(rt, min(gl, NROWS - 1)).y + G - 1) / (unsigned)G);
                int c8 = (int)((unsigned)(rt_get(rt, NROWS - 1).y + G - 1) / (unsigned)G);
                cv = max(cv, __builtin_amdgcn_update_dpp(0, cv, 0x128 /* row_ror:8 */, 0xf, 0xf, true));
                c8 = max(c8, __builtin_amdgcn_update_dpp(0, c8, 0x128, 0xf, 0xf, true));
                auto rows_max = [](int v) {
                    auto q = __builtin_amdgcn_permlane16_swap((unsigned)v, (unsigned)v, false, false);
                    v = max((int)q[0], (int)q[1]);
                    q = __builtin_amdgcn_permlane32_swap((unsigned)v, (unsigned)v, false, false);
                    return max((int)q[0], (int)q[1]);
                };
                cv = rows_max(cv);
                c8 = rows_max(c8);
#pragma unroll
                for (int r = 0; r < NROWS; r++)
                    trips_of[r] = (r == NROWS - 1) ? __builtin_amdgcn_readlane(c8, 0) : __builtin_amdgcn_readlane(cv, r);
            }
            unsigned D[NROWS];                                   // two bits per candidate of my share of each tile row
            int2 rv_next = rt_get(rt, 0);
#pragma unroll
            for (int r = 0; r < NROWS; r++) {
                const int c0 = rv_next.x, span = rv_next.y;
                if (r + 1 < NROWS) rv_next = rt_get(rt, r + 1);
                const int lim = (int)((unsigned)(span - gl + G - 1) >> LOG2G);   // my candidates: slots cb + k G, k < lim (may be <= 0)
                const int cb = c0 + gl;
                const int trips = trips_of[r];                    // <= 16 (host check: a tile row holds <= 16 G atoms)
                unsigned bits = 0;
                const float4 *cand = tile + cb;
                auto dist = [&](const float4 &q) {
                    const float dx = qi.x - q.x, dyy = qi.y - q.y, dzz = qi.z - q.z;
                    float t = __builtin_fmaf(dx, dx, nrl2);
                    t = __builtin_fmaf(dyy, dyy, t);
                    return __builtin_fmaf(dzz, dzz, t);
                };
                // rounding band (fp64 boxes): decided with the exact fp64 records (a far entry if listed)
                auto exact = [&](float &t, int gpj, int k) {
                    if (__builtin_fabsf(t) <= margin_v && k < lim) {
                        int kq = k;
                        asm volatile("" : "+s"(kq));
                        const int c = cb + kq * G;
                        int occ = oc;
                        asm volatile("" : "+v"(occ));
                        const int tcr = occ % BX + TX * (((occ / BX) % BY + r % 3) + TY * (occ / (BX * BY) + r / 3));
                        const int tc = tcr + (c >= T.off[tcr + 1] ? 1 : 0) + (c >= T.off[tcr + 2] ? 1 : 0);
                        const int sh = T.shift[tc];
                        const Rec<real> ri = a.rec[p], rj = a.rec[gpj];
                        const real ex = ri.x - (rj.x + (real)((sh & 3) - 1) * a.g.len[0]);
                        const real ey = ri.y - (rj.y + (real)(((sh >> 2) & 3) - 1) * a.g.len[1]);
                        const real ez = ri.z - (rj.z + (real)(((sh >> 4) & 3) - 1) * a.g.len[2]);
                        t = (ex * ex + ey * ey + ez * ez < a.rlist2) ? -1.f : 1.f;
                    }
                };
                auto shift_in = [&](float t) { bits = __builtin_amdgcn_alignbit(bits, __float_as_uint(t), 30); };   // bits = bits << 2 | t >> 30
#pragma clang loop unroll(disable) vectorize(disable) interleave(disable)
                for (int k = 0; k + 1 < trips; k += 2) {             // (reads past my share / the tile: harmless)
                    float4 q[2];
                    float t[2];
                    q[0] = cand[k * G]; q[1] = cand[(k + 1) * G];
                    if constexpr (!BAND) { asm volatile("" : : "v"(q[0].w)); asm volatile("" : : "v"(q[1].w)); }
                    t[0] = dist(q[0]); t[1] = dist(q[1]);
                    if constexpr (BAND) {
                        const float tm = __builtin_fminf(__builtin_fabsf(t[0]), __builtin_fabsf(t[1]));
                        if (__builtin_expect(__builtin_amdgcn_ballot_w64(tm <= margin_v) != 0, 0)) {
                            exact(t[0], __float_as_int(q[0].w), k);
                            exact(t[1], __float_as_int(q[1].w), k + 1);
                        }
                    }
                    shift_in(t[0]); shift_in(t[1]);
                }
                if (trips & 1) {
                    const float4 q = cand[(trips - 1) * G];
                    if constexpr (!BAND) asm volatile("" : : "v"(q.w));
                    float t = dist(q);
                    if constexpr (BAND) {
                        if (__builtin_expect(__builtin_amdgcn_ballot_w64(__builtin_fabsf(t) <= margin_v) != 0, 0))
                            exact(t, __float_as_int(q.w), trips - 1);
                    }
                    shift_in(t);
                }
                // candidate k sits at digit trips - 1 - k: what lies past my share (k >= lim) is the low trips - lim digits
                const int drop = 2 * (trips - max(lim, 0));
                bits = drop >= 32 ? 0u : ((bits >> drop) << drop);
                if (r == 4) {                                         // the atom itself (its cell is the middle one of row 4)
                    const int d = ti - c0;
                    if ((d & (G - 1)) == gl) bits &= ~(3u << (2 * (trips - 1 - (d >> LOG2G))));
                }
                D[r] = bits;
            }
            // ---- phase 2: near entries of all lanes first, far entries behind them --------------------------------------
            int mineN = 0, mineH = 0;
#pragma unroll
            for (int r = 0; r < NROWS; r++) {
                const unsigned H = (D[r] >> 1) & 0x55555555u;         // listed: the sign bit of t
                mineH += __popc(H);
                mineN += __popc(H & D[r]);                            // ... and |t| >= 2
            }
            const int mineF = mineH - mineN;
            auto group_prefix = [&](int v) {
                int t = __builtin_amdgcn_update_dpp(0, v, DPP_ROW_SHR1, 0xf, 0xf, true);
                v += gl >= 1 ? t : 0;
                t = __builtin_amdgcn_update_dpp(0, v, DPP_ROW_SHR2, 0xf, 0xf, true);
                v += gl >= 2 ? t : 0;
                t = __builtin_amdgcn_update_dpp(0, v, DPP_ROW_SHR4, 0xf, 0xf, true);
                v += gl >= 4 ? t : 0;
                return v;
            };
            const int inclN = group_prefix(mineN), inclF = group_prefix(mineF);
            const int totalN = __shfl(inclN, lane | (G - 1));
            unsigned short *epN = rowbuf + (unsigned)(inclN - mineN), *epF = rowbuf + (unsigned)(totalN + inclF - mineF);
            unsigned short *const ep_last = rowbuf + (ustride - 1u);
            // Round 4: TEN emission loops instead of eighteen.  The listed / near masks of a row have their digits at the EVEN bit
            // positions (digit j at bit 2 j), so the masks of two rows fit one word, the second shifted to the odd positions;
            // the pairs are OPPOSITE rows, (dy, dz) with (-dy, -dz) -- their hits add up to nearly the same number for every atom,
            // and a loop runs as long as the busiest of 64 lanes -- and row 4 stays alone.  Bit b of a word: row A (b even) or B
            // (b odd), digit b >> 1 = tile slot c0 + gl + (trips - 1 - j) G.
#pragma unroll
            for (int w = 0; w < 5; w++) {
                const int rA = w, rB = 8 - w;
                const unsigned HA = (D[rA] >> 1) & 0x55555555u;
                unsigned WN = HA & D[rA], WF = HA & ~D[rA];
                int baseA = (rt_get(rt, rA).x + gl + (trips_of[rA] - 1) * G) << a.idx_shift, dBA = 0;
                if (w < 4) {
                    const unsigned HB = (D[rB] >> 1) & 0x55555555u;
                    WN |= (HB & D[rB]) << 1;
                    WF |= (HB & ~D[rB]) << 1;
                    dBA = ((rt_get(rt, rB).x + gl + (trips_of[rB] - 1) * G) << a.idx_shift) - baseA;
                }
                asm volatile("" : "+v"(baseA), "+v"(dBA));
#ifdef EMDEE_BUILD_ABLATE      // timing experiments only (the lists are wrong): 2 no emission
                if (EMDEE_BUILD_ABLATE & 2) { WN = 0; WF = 0; }
                if (EMDEE_BUILD_ABLATE & 32) { WN |= WF; WF = 0; }     // 32: one class (all hits through the near loops)
#endif
                auto entry = [&](int b) {                              // base of the row the bit belongs to, minus the digit's step
                    // (a shift, not a multiply: v_mul_lo_u32 is a quarter-rate instruction, and the compiler picks it for a product)
                    return baseA + ((b & 1) ? dBA : 0) - ((b >> 1) << kshift);
                };
                while (WN) {
                    const int b = __ffs((int)WN) - 1;
                    WN &= WN - 1;
                    asm volatile("" : "+v"(WN));
                    *(epN < ep_last ? epN : ep_last) = (unsigned short)entry(b);
                    epN++;
                }
                while (WF) {
                    const int b = __ffs((int)WF) - 1;
                    WF &= WF - 1;
                    asm volatile("" : "+v"(WF));
                    *(epF < ep_last ? epF : ep_last) = (unsigned short)entry(b);
                    epF++;
                }
            }
            if (have && EMDEE_BOUND(BS_BUILD_ROW, p, a.n)) {
                constexpr int BLKL = EPL * GL;                // entries per lane-major block of the force kernels' rows
                // (left to itself the compiler interleaves two trips of this loop behind a run-time alias check and splits each
                // 16-byte store into four: 130 instructions per atom where 40 do -- the flush was 0.32 ms of the build for that)
                EMDEE_PLAIN_LOOP
                for (int c = gl * EPL; c < a.stride; c += G * EPL) {
                    const unsigned short *src = rowbuf + (c / BLKL) * BLKL + (c % BLKL) / EPL;   // entries src[GL t], t = 0..7
                    uint4 q;
                    q.x = (unsigned)src[0 * GL] | ((unsigned)src[1 * GL] << 16);
                    q.y = (unsigned)src[2 * GL] | ((unsigned)src[3 * GL] << 16);
                    q.z = (unsigned)src[4 * GL] | ((unsigned)src[5 * GL] << 16);
                    q.w = (unsigned)src[6 * GL] | ((unsigned)src[7 * GL] << 16);
                    *reinterpret_cast<uint4 *>(row + c) = q;
                }
                if (gl == G - 1) {                            // the last lane's inclusive prefixes add up to the row length
                    const unsigned total = (unsigned)(inclN + inclF);
                    a.cnt[p] = act ? (int)(min(total, ustride) | (a.far_skip ? (min((unsigned)inclN, ustride) << 8) : 0u)) : 0;
                    if (total > ustride) atomicMax(&a.flags[0], (int)total);
                }
            }
        }
        return;
    }
    if constexpr (ALG % 10 == 3 || ALG % 10 == 5) {   // ALG 13 / 15: the same with candidates dealt round-robin (below)
        // ALG 2 with a leaner candidate loop (the build is VALU-issue bound: 2.7 G wave-instructions per rebuild at
        // 10^7 atoms, profiles/r02): the trip count of a tile row is made WAVE-uniform (the longest chunk in the
        // wavefront; a lane whose chunk is shorter tests slots past its chunk and drops those bits afterwards), so
        // the loop needs no per-lane exit bookkeeping on the execution mask; the hit bit is shifted in by the carry
        // input of one add (v_cmp -> vcc, v_addc: bits = 2 bits + hit); and the squared distance is accumulated
        // starting from -r_list^2, so "surely inside" and "inside the rounding band" are compares against +-margin.
        static_assert(G == 4 || G == 8 || G == 16, "two-phase build: 4, 8 or 16 lanes per atom");
        // ALG 3: two 16-bit fields per word (a lane's chunk of a tile row holds <= 16 candidates); ALG 5: one 32-bit field per
        // word, for long cutoffs / dense boxes (rc = 3.5 sigma: 132 candidates per row, 17 per lane)
        constexpr int FIELD = ALG % 10 == 3 ? BUILD2_FIELD : 32, PER = 32 / FIELD;
        constexpr int NROWS = 9, NWORDS = (NROWS + PER - 1) / PER;
        constexpr bool BAND = sizeof(real) == 8;              // fp32 boxes: the fp32 test is the definition of the set
#ifndef EMDEE_BUILD_UNROLL
#define EMDEE_BUILD_UNROLL 2
#endif
        constexpr int UNR = EMDEE_BUILD_UNROLL;
        float nrl2 = -rl2, nmargin_v = BAND ? -a.margin : 0.f, margin_v = a.margin;
        asm volatile("" : "+v"(nrl2), "+v"(nmargin_v), "+v"(margin_v));   // loop-invariant operands stay in VGPRs
#ifndef EMDEE_BUILD_ROWTAB
#define EMDEE_BUILD_ROWTAB 1
#endif
        // The build is bound by VALU issue, and a third of its instructions were per-row bookkeeping (profiles/r02): the
        // candidate rows of an atom come from the brick's row table (one ds_read_b64 each, immediate offsets), and the
        // wave-uniform trip counts of all 9 rows are found at once -- lane gl of every group holds the chunk length of
        // row gl, three max steps combine the groups of the wavefront, 9 v_readlane move the result to scalars.
        constexpr bool RT = EMDEE_BUILD_ROWTAB != 0;
        // an odd trip count ends with a single-candidate step instead of being rounded up (row spans of ~53 slots over 8
        // lanes give 7 trips: rounding to 8 tested 14 % more slots than there are)
        constexpr bool TAIL = RT && UNR == 2;
#ifndef EMDEE_BUILD_STRIDED
#define EMDEE_BUILD_STRIDED 1
#endif
        // Candidates are dealt to the lanes of a group round-robin (lane gl tests slots c0 + gl, c0 + gl + G, ...), not in
        // contiguous chunks: in-range candidates come in runs along a tile row, so contiguous chunks gave some lanes all of
        // a row's hits and others none -- and the emission loops below run as long as the busiest lane of the wavefront
        // (measured: emission 0.92 ms of a 2.87 ms build, more than the distance tests).  The 8 lanes of a group also read
        // 128 contiguous bytes per step.
        // The rows then list slots G apart in consecutive entries: fine for the force kernels that read 8-byte plane values
        // or 16-byte records with 4 lanes per atom, an 8-way LDS bank conflict for 32-byte records read by 8 lanes (measured
        // on the rc = 3.5 mixture: build 8.1 -> 6.6 ms, force 4.1 -> 6.0 ms) -- the host picks ALG 13/15 only for the former.
        constexpr bool STRIDED = RT && EMDEE_BUILD_STRIDED != 0 && ALG >= 10;
#ifndef EMDEE_BUILD_PAIR_OPPOSITE
#define EMDEE_BUILD_PAIR_OPPOSITE 1
#endif
        // Which two tile rows share a 32-bit word of hit bits (16-bit fields).  The emission loop of a word runs as long as
        // the busiest of the 64 lanes has hits in it, and an atom near a face of its cell has many hits in the row beyond
        // that face and few in the opposite one: rows (dy, dz) and (-dy, -dz) in one word -- (0,8) (1,7) (2,6) (3,5) (4) --
        // have a nearly constant sum where neighbouring rows (0,1) (2,3) ... do not.
        constexpr bool OPP = EMDEE_BUILD_PAIR_OPPOSITE != 0 && PER == 2;
        auto row_word = [](int r) constexpr { return OPP ? (r <= 4 ? r : 8 - r) : r / PER; };
        auto row_half = [](int r) constexpr { return OPP ? (r > 4 ? 1 : 0) : r % PER; };
        constexpr int LOG2G = G == 4 ? 2 : (G == 8 ? 3 : 4), KSTEP = STRIDED ? G : 1;
        // first own atom of this wavefront in a round (a scalar): a wavefront whose eight groups all lie past the brick's last
        // atom in the last round has nothing to list -- it used to walk the whole round with empty rows (~450 instructions of
        // bookkeeping: 4 of the 40 wavefront-rounds of a 282-atom brick)
        const int wave_first = __builtin_amdgcn_readfirstlane((tid / WAVE) * (WAVE / G));
        for (int ob = 0; ob < n_own; ob += NGROUPS) {         // wave-uniform trip count
            if (ob + wave_first >= n_own) break;
            const int o = ob + gid;
            const bool have = o < n_own;
            // own atom: cell-order slot, tile slot, own cell and "owned" flag, located once while the tile was staged
            const int2 info = have ? T.oinfo[o] : make_int2(0, 0);
            const int ti = info.y & 0xffff, p = info.x, oc = info.y >> 20;
            const bool act = have && ((info.y >> 16) & 1) != 0;   // ghosts own no row
            const float4 qi = tile[ti];
            unsigned short *row = a.nbr + (size_t)p * a.stride;
            for (int c = gl * EPL; c < a.stride; c += G * EPL) *reinterpret_cast<uint4 *>(rowbuf + c) = fill;
            unsigned word[NWORDS];
            int cbase[NROWS];
            const unsigned *rt = rtab + (act ? oc * NSUB + ((info.y >> 17) & 3) : Shape::NOC * NSUB) * 9;
            int2 rv_next = make_int2(0, 0);
            int trips_of[NROWS];
            if constexpr (RT) {
                rv_next = rt_get(rt, 0);
                // chunk length of row gl (G = 8: rows 0..7 in the lanes, row 8 apart; G = 16: lanes 0..8 hold all nine; G = 4: rows
                // 0..3 in cv, 4..7 in cw, row 8 apart)
                auto chunk_of = [&](int r) { return (int)((unsigned)(rt_get(rt, r).y + G - 1) / (unsigned)G); };
                int cv = chunk_of(min(gl, NROWS - 1));
                int cw = G == 4 ? chunk_of(4 + gl) : 0;
                int c8 = chunk_of(NROWS - 1);
                // across the groups of a 16-lane row with row rotations, across the four rows with the gfx950 row / half swaps
                // (no LDS, no address registers)
                auto ror4 = [](int v) { return max(v, __builtin_amdgcn_update_dpp(0, v, 0x124 /* row_ror:4 */, 0xf, 0xf, true)); };
                auto ror8 = [](int v) { return max(v, __builtin_amdgcn_update_dpp(0, v, 0x128 /* row_ror:8 */, 0xf, 0xf, true)); };
                if constexpr (G == 4) { cv = ror4(cv); cw = ror4(cw); c8 = ror4(c8); }
                if constexpr (G <= 8) {
                    cv = ror8(cv); c8 = ror8(c8);
                    if constexpr (G == 4) cw = ror8(cw);
                }
                auto rows_max = [](int v) {
                    auto q = __builtin_amdgcn_permlane16_swap((unsigned)v, (unsigned)v, false, false);
                    v = max((int)q[0], (int)q[1]);
                    q = __builtin_amdgcn_permlane32_swap((unsigned)v, (unsigned)v, false, false);
                    return max((int)q[0], (int)q[1]);
                };
                cv = rows_max(cv);
                if constexpr (G <= 8) c8 = rows_max(c8);
                if constexpr (G == 4) cw = rows_max(cw);
#pragma unroll
                for (int r = 0; r < NROWS; r++) {
                    int m;
                    if (G <= 8 && r == NROWS - 1) m = __builtin_amdgcn_readlane(c8, 0);
                    else if (G == 4 && r >= 4) m = __builtin_amdgcn_readlane(cw, r - 4);
                    else m = __builtin_amdgcn_readlane(cv, r);
                    trips_of[r] = TAIL ? m : ((m + UNR - 1) & ~(UNR - 1));
                    // (a lane's share of a tile row must fit its bit field: the host picked the field from the widest 3-cell run,
                    // which the x sub-bins undercut -- and with 4 lanes per atom only THEY keep a row within 16 x 4 slots)
                    if (FIELD < 32 && m > FIELD && lane == 0) atomicMax(&a.flags[4], m);
                }
            }
#pragma unroll
            for (int r = 0; r < NROWS; r++) {
                int tcr = 0, c0, span;
                if constexpr (RT) {
                    c0 = rv_next.x; span = rv_next.y;
                    if (r + 1 < NROWS) rv_next = rt_get(rt, r + 1);                        // one row ahead of its use
                } else {
                    const int ox = oc % BX, oy = (oc / BX) % BY, oz = oc / (BX * BY);
                    tcr = ox + TX * ((oy + r % 3) + TY * (oz + r / 3));           // cell x-1 of tile row (dy, dz) = (r%3-1, r/3-1)
                    c0 = T.off[tcr];
                    span = act ? T.off[tcr + 3] - c0 : 0;                           // cells x-1, x, x+1: contiguous
                }
                const int chunk = (int)((unsigned)(span + G - 1) / (unsigned)G);    // <= BUILD2_FIELD (host check)
                const int first = STRIDED ? gl : (int)__umul24((unsigned)gl, (unsigned)chunk);
                // my candidates (may be <= 0 in the chunked form): slots cb + k KSTEP, k < lim
                const int lim = STRIDED ? (int)((unsigned)(span - gl + G - 1) >> LOG2G) : min(chunk, span - first);
                const int cb = c0 + first;
                cbase[r] = cb - row_half(r) * FIELD * KSTEP;
                // scalar (chunk is the same in all lanes of a group); unrolled UNR times
#ifdef EMDEE_BUILD_ABLATE      // timing experiments only (the lists are wrong): 1 no candidate loop, 2 no emission, 4 no flush
                const int trips = (EMDEE_BUILD_ABLATE & 1) ? 0 : (RT ? trips_of[r] : ((wave_group_max<G>(chunk) + UNR - 1) & ~(UNR - 1)));
#else
                const int trips = RT ? trips_of[r] : ((wave_group_max<G>(chunk) + UNR - 1) & ~(UNR - 1));
#endif
                unsigned bits = 0;
                const float4 *cand = tile + cb;
                // d^2 - r_list^2 of candidate q, accumulated from -r_list^2
                auto dist = [&](const float4 &q) {
                    const float dx = qi.x - q.x, dyy = qi.y - q.y, dzz = qi.z - q.z;
                    float t = __builtin_fmaf(dx, dx, nrl2);
                    t = __builtin_fmaf(dyy, dyy, t);
                    return __builtin_fmaf(dzz, dzz, t);
                };
                // rounding band (fp64 boxes): decided with the exact fp64 records, for the lanes concerned -- and only for
                // slots of the lane's own chunk: what lies past it is dropped below and may not be a record
                auto exact = [&](float &t, int gpj, int k) {
                    if (__builtin_fabsf(t) <= margin_v && k < lim) {
                        int kq = k;
                        asm volatile("" : "+s"(kq));                         // (worked out here, not carried through the loop)
                        const int c = cb + kq * KSTEP;
                        if constexpr (RT) {                                  // rare path: the tile row is worked out here, not per row
                            int occ = oc;
                            asm volatile("" : "+v"(occ));
                            tcr = occ % BX + TX * (((occ / BX) % BY + r % 3) + TY * (occ / (BX * BY) + r / 3));
                        }
                        const int tc = tcr + (c >= T.off[tcr + 1] ? 1 : 0) + (c >= T.off[tcr + 2] ? 1 : 0);
                        const int sh = T.shift[tc];
                        const Rec<real> ri = a.rec[p], rj = a.rec[gpj];
                        const real ex = ri.x - (rj.x + (real)((sh & 3) - 1) * a.g.len[0]);
                        const real ey = ri.y - (rj.y + (real)(((sh >> 2) & 3) - 1) * a.g.len[1]);
                        const real ez = ri.z - (rj.z + (real)(((sh >> 4) & 3) - 1) * a.g.len[2]);
                        t = (ex * ex + ey * ey + ez * ez < a.rlist2) ? -1.f : 1.f;
                    }
                };
                // the hit bit enters through the carry: bits = 2 bits + (t < -margin)
                // (fp32 boxes, margin 0: the sign of the fused sum IS the definition of the listed set)
                // (round 3: ONE instruction.  A candidate whose |t| is inside the margin has been settled by exact(), which
                // leaves t = -1 or +1; for every other candidate t < -margin is t < 0; and t = d^2 - r_list^2 is never -0.
                // So "listed" is the sign bit of t, and v_alignbit_b32 {bits, t} >> 31 is bits = 2 bits + sign(t) --
                // what the compare + add-with-carry pair did in two: -6 % instructions in the candidate loop.)
                auto shift_in = [&](float t) {
#ifdef EMDEE_BUILD_CMP_ADDC
                    asm volatile("v_cmp_lt_f32_e32 vcc, %1, %2\n\tv_addc_co_u32_e32 %0, vcc, %0, %0, vcc"
                                 : "+v"(bits) : "v"(t), "v"(nmargin_v) : "vcc");
#else
                    bits = __builtin_amdgcn_alignbit(bits, __float_as_uint(t), 31);
#endif
                };
                // UNR candidates per trip, loaded at its top: with six wavefronts per SIMD the LDS latency hides behind the
                // other waves' arithmetic, and nothing is carried from trip to trip (no register rotation).  The band test
                // of the group is one compare of the smallest |t| with the margin.
#pragma clang loop unroll(disable) vectorize(disable) interleave(disable)
                for (int k = 0; k + (TAIL ? 1 : 0) < trips; k += UNR) {             // (reads past the chunk / the tile: harmless)
                    float4 q[UNR];
                    float t[UNR];
#pragma unroll
                    for (int u = 0; u < UNR; u++) q[u] = cand[(k + u) * KSTEP];
#ifdef EMDEE_BUILD_ABLATE      // 8: half the candidate reads (the second candidate of a trip is the first again); 16: every read twice
                    if (EMDEE_BUILD_ABLATE & 8) { q[1] = q[0]; asm volatile("" : "+v"(q[1].x)); }
                    if (EMDEE_BUILD_ABLATE & 16) {
                        float4 extra = cand[(k + 1) * KSTEP + 1];
                        asm volatile("" : : "v"(extra.x), "v"(extra.y), "v"(extra.z), "v"(extra.w));
                        extra = cand[k * KSTEP + 1];
                        asm volatile("" : : "v"(extra.x), "v"(extra.y), "v"(extra.z), "v"(extra.w));
                    }
#endif
                    // (keeps the whole 16-byte records alive: a ds_read_b96 costs 8 LDS cycles per wavefront, a ds_read_b128 4)
                    if constexpr (!BAND) {
#pragma unroll
                        for (int u = 0; u < UNR; u++) asm volatile("" : : "v"(q[u].w));
                    }
#pragma unroll
                    for (int u = 0; u < UNR; u++) t[u] = dist(q[u]);
                    if constexpr (BAND) {
                        float tm = __builtin_fabsf(t[0]);
#pragma unroll
                        for (int u = 1; u < UNR; u++) tm = __builtin_fminf(tm, __builtin_fabsf(t[u]));
                        if (__builtin_expect(__builtin_amdgcn_ballot_w64(tm <= margin_v) != 0, 0)) {
#pragma unroll
                            for (int u = 0; u < UNR; u++) exact(t[u], __float_as_int(q[u].w), k + u);
                        }
                    }
#pragma unroll
                    for (int u = 0; u < UNR; u++) shift_in(t[u]);
                }
                if (TAIL && (trips & 1)) {
                    const float4 q = cand[(trips - 1) * KSTEP];
                    asm volatile("" : : "v"(q.w));          // (the whole record: a ds_read_b96 costs twice the LDS cycles of a ds_read_b128)
                    float t = dist(q);
                    if constexpr (BAND) {
                        if (__builtin_expect(__builtin_amdgcn_ballot_w64(__builtin_fabsf(t) <= margin_v) != 0, 0))
                            exact(t, __float_as_int(q.w), trips - 1);
                    }
                    shift_in(t);
                }
                // candidate k sits at bit trips-1-k: reverse, drop what lies past my chunk
                if constexpr (RT && FIELD < 32)                                     // one v_bfe_u32: bits [32-trips, 32-trips+lim)
                    bits = __builtin_amdgcn_ubfe(__builtin_bitreverse32(bits), (unsigned)(32 - trips), (unsigned)max(lim, 0));
                else
                    bits = lim > 0 ? ((__builtin_bitreverse32(bits) >> (32 - trips)) & (lim >= 32 ? ~0u : ((1u << lim) - 1u))) : 0u;
                if (r == 4) {                                                       // the atom itself
                    if constexpr (STRIDED) {
                        const int d = ti - c0;                                      // >= 0: the atom's cell is the middle one of row 4
                        if ((d & (G - 1)) == gl) bits &= ~(1u << (d >> LOG2G));
                    } else {
                        const int ks = ti - cb;
                        if (ks >= 0 && ks < lim) bits &= ~(1u << ks);
                    }
                }
                if (row_half(r)) word[row_word(r)] |= bits << (row_half(r) * FIELD);
                else word[row_word(r)] = bits;
            }
            // ---- phase 2 (as ALG 2): prefix over the lanes of the group, then every lane emits its hits --------
            int mine = 0;
#pragma unroll
            for (int w = 0; w < NWORDS; w++) mine += __popc(word[w]);
            int incl = mine;
            {
                int t = __builtin_amdgcn_update_dpp(0, incl, DPP_ROW_SHR1, 0xf, 0xf, true);
                incl += gl >= 1 ? t : 0;
                t = __builtin_amdgcn_update_dpp(0, incl, DPP_ROW_SHR2, 0xf, 0xf, true);
                incl += gl >= 2 ? t : 0;
                if (G >= 8) {
                    t = __builtin_amdgcn_update_dpp(0, incl, DPP_ROW_SHR4, 0xf, 0xf, true);
                    incl += gl >= 4 ? t : 0;
                }
                if (G == 16) {
                    t = __builtin_amdgcn_update_dpp(0, incl, DPP_ROW_SHR8, 0xf, 0xf, true);
                    incl += gl >= 8 ? t : 0;
                }
            }
            // Hits go to the row buffer in PLAIN order (hit number e at rowbuf[e]: one shifted add and one 2-byte LDS store
            // per hit); the lane-major block layout the force kernels read (row_position) is produced while the row is
            // flushed, each 16-byte output chunk gathering its 8 entries.  A row that overflows the stride keeps
            // overwriting its last slot: the host sees the count, grows the stride and builds again.
            // (a running pointer clamped to the row's last slot; the opaque use of W keeps the loop from being rewritten as a
            // counted one, which costs a decrement and a compare per hit instead of the compare with zero)
            unsigned short *ep = rowbuf + (unsigned)(incl - mine), *const ep_last = rowbuf + (ustride - 1u);
#pragma unroll
            for (int w = 0; w < NWORDS; w++) {
                unsigned W = word[w];
#ifdef EMDEE_BUILD_ABLATE
                if (EMDEE_BUILD_ABLATE & 2) W = 0;
#endif
                int cA = cbase[OPP ? w : PER * w] << a.idx_shift,
                    cB = (OPP ? (w < 4 ? cbase[8 - w] : 0) : ((PER == 2 && 2 * w + 1 < NROWS) ? cbase[2 * w + 1] : 0)) << a.idx_shift;
                asm volatile("" : "+v"(cA), "+v"(cB));                             // shifted once, here
                const int kshift = a.idx_shift + (STRIDED ? LOG2G : 0);          // bit k of a field is slot cb + k KSTEP
                while (W) {
                    const int k = __ffs((int)W) - 1;
                    W &= W - 1;
                    asm volatile("" : "+v"(W));
                    *(ep < ep_last ? ep : ep_last) = (unsigned short)((k << kshift) + ((PER == 2 && k >= FIELD) ? cB : cA));
                    ep++;
                }
            }
#ifdef EMDEE_BUILD_ABLATE
            if (have && !(EMDEE_BUILD_ABLATE & 4)) {
#else
            if (have && EMDEE_BOUND(BS_BUILD_ROW, p, a.n)) {
#endif
                constexpr int BLKL = EPL * GL;                // entries per lane-major block of the force kernels' rows
                EMDEE_PLAIN_LOOP
                for (int c = gl * EPL; c < a.stride; c += G * EPL) {
                    const unsigned short *src = rowbuf + (c / BLKL) * BLKL + (c % BLKL) / EPL;   // entries src[GL t], t = 0..7
                    uint4 q;
                    q.x = (unsigned)src[0 * GL] | ((unsigned)src[1 * GL] << 16);
                    q.y = (unsigned)src[2 * GL] | ((unsigned)src[3 * GL] << 16);
                    q.z = (unsigned)src[4 * GL] | ((unsigned)src[5 * GL] << 16);
                    q.w = (unsigned)src[6 * GL] | ((unsigned)src[7 * GL] << 16);
                    *reinterpret_cast<uint4 *>(row + c) = q;
                }
                if (gl == G - 1) {                            // the last lane's inclusive prefix is the row length
                    a.cnt[p] = act ? (int)min((unsigned)incl, ustride) : 0;
                    if ((unsigned)incl > ustride) atomicMax(&a.flags[0], incl);
                }
            }
        }
        return;
    }
    if constexpr (ALG == 2) {
        static_assert(G == 8 || G == 16, "two-phase build: 8 or 16 lanes per atom");
        constexpr int NROWS = 9, NWORDS = (NROWS + 1) / 2;
        for (int ob = 0; ob < n_own; ob += NGROUPS) {         // wave-uniform trip count
            const int o = ob + gid;
            const bool have = o < n_own;
            int ti = 0, p = 0, oc = 0;
            if (have) oc = brick_locate(T, o, ti, p);
            const bool act = have && ((T.oinfo[o].y >> 16) & 1) != 0;   // ghosts own no row
            const int ox = oc % BX, oy = (oc / BX) % BY, oz = oc / (BX * BY);
            const float4 qi = tile[ti];
            unsigned short *row = a.nbr + (size_t)p * a.stride;
            for (int c = gl * EPL; c < a.stride; c += G * EPL) *reinterpret_cast<uint4 *>(rowbuf + c) = fill;
            // ---- phase 1: one bit per candidate of my chunk of each of the 9 tile rows -------------
            unsigned word[NWORDS];
            int cbase[NROWS];
#pragma unroll
            for (int r = 0; r < NROWS; r++) {
                const int dy = r % 3 - 1, dz = r / 3 - 1;
                const int tcr = ox + TX * ((oy + 1 + dy) + TY * (oz + 1 + dz));   // cell x-1 of that tile row
                const int c0 = T.off[tcr];
                const int span = act ? T.off[tcr + 3] - c0 : 0;                    // cells x-1, x, x+1: contiguous
                const int chunk = (span + G - 1) / G;                               // <= BUILD2_FIELD (host check)
                const int first = gl * chunk;
                const int lim = min(chunk, span - first);                           // my candidates; may be <= 0
                const int cb = c0 + first;
                cbase[r] = (r & 1) ? cb - BUILD2_FIELD : cb;
                unsigned bits = 0;
                float4 qc = tile[cb];                                               // one candidate ahead
#pragma clang loop unroll(disable) vectorize(disable) interleave(disable)
                for (int k = 0; k < lim; k++) {                                     // divergent trip count
                    const float4 qn = tile[cb + k + 1];                             // (past the chunk: harmless)
                    bits |= (in_range_q(qi, qc, p, cb + k, tcr) ? 1u : 0u) << k;
                    qc = qn;
                }
                if (r == 4) {                                                       // the atom itself
                    const int ks = ti - cb;
                    if (ks >= 0 && ks < lim) bits &= ~(1u << ks);
                }
                if (r & 1) word[r / 2] |= bits << BUILD2_FIELD;
                else word[r / 2] = bits;
            }
            // ---- phase 2: prefix over the lanes of the group, then every lane emits its hits --------
            int mine = 0;
#pragma unroll
            for (int w = 0; w < NWORDS; w++) mine += __popc(word[w]);
            int incl = mine;
            {
                int t = __builtin_amdgcn_update_dpp(0, incl, DPP_ROW_SHR1, 0xf, 0xf, true);
                incl += gl >= 1 ? t : 0;
                t = __builtin_amdgcn_update_dpp(0, incl, DPP_ROW_SHR2, 0xf, 0xf, true);
                incl += gl >= 2 ? t : 0;
                t = __builtin_amdgcn_update_dpp(0, incl, DPP_ROW_SHR4, 0xf, 0xf, true);
                incl += gl >= 4 ? t : 0;
                if (G == 16) {
                    t = __builtin_amdgcn_update_dpp(0, incl, DPP_ROW_SHR8, 0xf, 0xf, true);
                    incl += gl >= 8 ? t : 0;
                }
            }
            unsigned e = (unsigned)(incl - mine);
#pragma unroll
            for (int w = 0; w < NWORDS; w++) {
                unsigned W = word[w];
                const int cA = cbase[2 * w], cB = (2 * w + 1 < NROWS) ? cbase[2 * w + 1] : 0;
                while (W) {
                    const int k = __ffs((int)W) - 1;
                    W &= W - 1;
                    const int c = k + (k >= BUILD2_FIELD ? cB : cA);
                    if (e < ustride && EMDEE_BOUND(BS_BUILD_ROWBUF, row_position<GL>(e), ustride)) rowbuf[row_position<GL>(e)] = (unsigned short)(c << a.idx_shift);
                    e++;
                }
            }
            if (have && EMDEE_BOUND(BS_BUILD_ROW, p, a.n)) {
                for (int c = gl * EPL; c < a.stride; c += G * EPL)
                    *reinterpret_cast<uint4 *>(row + c) = *reinterpret_cast<const uint4 *>(rowbuf + c);
                if (gl == G - 1) {                            // the last lane's inclusive prefix is the row length
                    a.cnt[p] = act ? (int)min((unsigned)incl, ustride) : 0;
                    if ((unsigned)incl > ustride) atomicMax(&a.flags[0], incl);
                }
            }
        }
        return;
    }
    for (int ob = 0; ob < n_own; ob += NGROUPS) {             // wave-uniform trip count
        const int o = ob + gid;
        const bool have = o < n_own;
        int ti = 0, p = 0, oc = 0;
        if (have) oc = brick_locate(T, o, ti, p);
        const bool act = have && ((T.oinfo[o].y >> 16) & 1) != 0;   // ghosts own no row
        const int ox = oc % BX, oy = (oc / BX) % BY, oz = oc / (BX * BY);
        const float4 qi = tile[ti];
        unsigned short *row = a.nbr + (size_t)p * a.stride;
        for (int c = gl * EPL; c < a.stride; c += G * EPL) *reinterpret_cast<uint4 *>(rowbuf + c) = fill;
        unsigned count = 0;
#pragma unroll 1
        for (int dz = -1; dz <= 1; dz++) {
#pragma unroll 1
            for (int dy = -1; dy <= 1; dy++) {
                const int tcr = ox + TX * ((oy + 1 + dy) + TY * (oz + 1 + dz));   // cell x-1 of that tile row
                const int c0 = T.off[tcr];
                const int span = act ? T.off[tcr + 3] - c0 : 0;                    // cells x-1, x, x+1: contiguous
                const int wspan = wave_group_max<G>(span);
                for (int cb = 0; cb < wspan; cb += G) {
                    const int c = c0 + cb + gl;
                    bool pass = false;
                    if (cb + gl < span && c != ti) pass = in_range(qi, p, c, tcr);
                    const unsigned long long mask = __ballot(pass);
                    const unsigned half = upper ? (unsigned)(mask >> 32) : (unsigned)mask;
                    const unsigned bits = (half >> gshift) & gmask;
                    if (pass) {
                        const unsigned e = count + __popc(bits & ltmask);
                        if (e < ustride && EMDEE_BOUND(BS_BUILD_ROWBUF, row_position<GL>(e), ustride)) rowbuf[row_position<GL>(e)] = (unsigned short)(c << a.idx_shift);
                    }
                    count += __popc(bits);
                }
            }
        }
        if (have && EMDEE_BOUND(BS_BUILD_ROW, p, a.n)) {
            for (int c = gl * EPL; c < a.stride; c += G * EPL)
                *reinterpret_cast<uint4 *>(row + c) = *reinterpret_cast<const uint4 *>(rowbuf + c);
            if (gl == 0) {
                a.cnt[p] = act ? (int)min(count, ustride) : 0;
                if (count > ustride) atomicMax(&a.flags[0], (int)count);
            }
        }
    }
}

// ------------------------------------------------------------------------------------ force / stats
template <typename real, class Shape, int THREADS, int G, int MODE, int BITMASK, bool UNI = false>
__global__ __launch_bounds__(THREADS) void k_brick(BrickArgs<real> a) {
    constexpr int NGROUPS = (THREADS / WAVE) * (WAVE / G);
    constexpr int BLK = EPL * G;

    // All LDS lives in the dynamic region with 16-byte carve offsets (a static __shared__ in front
    // would shift the base and put the ds_read_b128 gathers off their natural alignment).
    extern __shared__ __attribute__((aligned(16))) unsigned char s_dyn[];
    constexpr bool SOA = UNI && MODE != BRICK_STATS;   // coordinate planes only (see SOA_SLOTS)
    // force-only single-species kernels (the MD loop): coordinates scaled by 1/sigma in the tile, constants folded (LJUni)
    constexpr bool FAST = SOA && BITMASK == EMDEE_FORCES;
    Rec<real> *tile = reinterpret_cast<Rec<real> *>(s_dyn);
    real *plane = reinterpret_cast<real *>(s_dyn);                       // SOA: x | y | z, soa_pitch() records apart
    constexpr int PITCH = soa_pitch<real>();
    const size_t tile_bytes = SOA ? (((size_t)3 * PITCH * sizeof(real) + 15) & ~(size_t)15) : (size_t)a.tile_cap * sizeof(Rec<real>);
    const size_t te_bytes = (sizeof(real) == 4 && !SOA) ? (((size_t)a.tile_cap * 4 + 15) & ~(size_t)15) : 0;
    float *tile_te = reinterpret_cast<float *>(s_dyn + tile_bytes);   // fp32 only
    BrickTables<Shape, THREADS> T;
    T.carve(s_dyn + tile_bytes + te_bytes);
    if (MODE == BRICK_STEP && a.guard != nullptr && *a.guard != 0) {
        // The host queued this step before it could know that the previous one moved an atom past skin/2:
        // the list is stale, so this launch (and, through the propagated flag, every later one of the
        // batch) leaves the state untouched.  The previous kernel has completed: the word is final.
        if (threadIdx.x == 0) *a.trigger = 1;
        return;
    }
    int bxi, byi, bzi, tile_n, n_own;
    if (!brick_setup<real, Shape, THREADS>(a, T, bxi, byi, bzi, tile_n, n_own)) return;
    const int tid = threadIdx.x, lane = tid & (WAVE - 1);
    // (far class skipped: rows end at their near entries while nobody has moved delta / 2; the statistics always see whole rows)
#ifdef EMDEE_EXPERIMENTS
    const bool far_skip = a.far_skip != 0;
#else
    constexpr bool far_skip = false;                          // (the product library has plain rows only)
#endif
    const bool near_only = far_skip && MODE != BRICK_STATS && *a.far_word == 0;
    auto row_len = [&](int c) { return far_skip ? (near_only ? (c >> 8) : (c & 255)) : c; };

    // ---- own-atom table entries first: their global loads fly while the tile is being staged --------
    // row length = cnt[p]; the build kernel wrote 0 for ghosts (they own no row and receive no force)
    int own_p[OWN_REGS], own_ti[OWN_REGS], own_m[OWN_REGS];
#pragma unroll
    for (int k = 0; k < OWN_REGS; k++) {
        const int o = tid + k * THREADS;
        own_p[k] = own_ti[k] = own_m[k] = 0;
        if (o < n_own) {
            brick_locate(T, o, own_ti[k], own_p[k]);
            own_m[k] = row_len(a.cnt[own_p[k]]);
        }
    }
    // ---- stage the tile: HBM -> LDS, unit stride inside each tile row, image shift applied -----------
    // fp32 boxes: the tile holds coordinates RELATIVE TO THE BRICK ORIGIN, worked out in fp64 with the +-L image shift folded
    // into the origin (as k_brick_build does for fp64 boxes): one rounding at the ulp of a brick-sized number (<= 1e-6 sigma)
    // instead of an addition of +-L to an absolute fp32 coordinate, whose result carries the ulp of [L, 2L) -- 1.5e-5 sigma in
    // the 10^7-atom box, and the reason the O(N) path used to miss the reference's own fp32 bound (test/runtests.jl:39-41).
    constexpr bool REL = sizeof(real) == 4;
    const bool refmath = REL && MODE == BRICK_FORCE && !SOA && a.refmath != 0;
    // cell-relative records: one integer per tile cell and dimension (rel_cell_const), worked out once per workgroup
    __shared__ float s_relc[REL ? Shape::TX + Shape::TY + Shape::TZ : 1];
    if (REL && a.rel) {
        rel_fill_consts<real, Shape>(a, bxi, byi, bzi, s_relc);
        __syncthreads();
    }
    double org[3] = {0.0, 0.0, 0.0};
    if (REL) {
        org[0] = (double)a.g.lo[0] + (double)(bxi * Shape::BX) * ((double)a.g.len[0] / (double)a.g.M[0]);
        org[1] = (double)a.g.lo[1] + (double)(byi * Shape::BY) * ((double)a.g.len[1] / (double)a.g.M[1]);
        org[2] = (double)a.g.lo[2] + (double)(bzi * Shape::BZ) * ((double)a.g.len[2] / (double)a.g.M[2]);
    }
    brick_for_each_slot(T, [&](int s, int tc, int tx, int ty, int tz) {
        const int gp = T.gbeg[tc] + (s - T.off[tc]);
        const int sh = T.shift[tc];
        Rec<real> r = a.rec[gp];
        if (refmath) {           // scaled positions, as the reference keeps them (src/nonbonded.jl:52-61,124)
            const size_t i = (size_t)a.perm[gp];
            r.x = a.user_pos[3 * i] / a.g.len[0]; r.y = a.user_pos[3 * i + 1] / a.g.len[1]; r.z = a.user_pos[3 * i + 2] / a.g.len[2];
        } else if (REL && a.rel) {                            // cell-relative records (BrickArgs::rel): fixed-point tile coordinates
            r.x = (real)rel_tile(r.x, s_relc[tx]); r.y = (real)rel_tile(r.y, s_relc[Shape::TX + ty]);
            r.z = (real)rel_tile(r.z, s_relc[Shape::TX + Shape::TY + tz]);
        } else if (REL) {
            r.x = (real)(((double)r.x + (double)((sh & 3) - 1) * (double)a.g.len[0]) - org[0]);
            r.y = (real)(((double)r.y + (double)(((sh >> 2) & 3) - 1) * (double)a.g.len[1]) - org[1]);
            r.z = (real)(((double)r.z + (double)(((sh >> 4) & 3) - 1) * (double)a.g.len[2]) - org[2]);
        } else if (sh != (1 | (1 << 2) | (1 << 4))) {   // (only tiles at a periodic face hold shifted cells: the rest skips the decode)
            r.x += (real)((sh & 3) - 1) * a.g.len[0];
            r.y += (real)(((sh >> 2) & 3) - 1) * a.g.len[1];
            r.z += (real)(((sh >> 4) & 3) - 1) * a.g.len[2];
        }
        if (FAST && a.uni.inv_sigma != (real)1) { r.x *= a.uni.inv_sigma; r.y *= a.uni.inv_sigma; r.z *= a.uni.inv_sigma; }
        if (!EMDEE_BOUND(BS_FORCE_TILE, s, SOA ? SOA_SLOTS : a.tile_cap)) return;
        if (SOA) { plane[s] = r.x; plane[PITCH + s] = r.y; plane[2 * PITCH + s] = r.z; }
        else tile[s] = r;
        if (sizeof(real) == 4 && !SOA) tile_te[s] = a.te[gp];
    });
    if (tid == 0) {   // the sentinel record every unused row entry points at: fails r2 < rc2, never NaN
        Rec<real> far;
        // (reference arithmetic wraps every difference into the box: there the sentinel is NaN, which fails r2 < rc2 as well)
        const real big = refmath ? (real)__builtin_nanf("") : (sizeof(real) == 8 ? (real)1e30 : (real)1e18);
        far.x = far.y = far.z = big; far.hs = 0;
        if (SOA) { plane[0] = big; plane[PITCH] = big; plane[2 * PITCH] = big; }
        else tile[0] = far;
        if (sizeof(real) == 4 && !SOA) tile_te[0] = 0.f;
    }
#pragma unroll
    for (int k = 0; k < OWN_REGS; k++) {
        const int o = tid + k * THREADS;
        if (o < n_own && EMDEE_BOUND(BS_FORCE_OWN, o, a.own_cap)) T.oinfo[o] = make_int2(own_p[k], (own_m[k] << 16) | own_ti[k]);
    }
    for (int o = tid + OWN_REGS * THREADS; o < n_own; o += THREADS) {   // very dense bricks only
        int ti, p;
        brick_locate(T, o, ti, p);
        if (EMDEE_BOUND(BS_FORCE_OWN, o, a.own_cap)) T.oinfo[o] = make_int2(p, (row_len(a.cnt[p]) << 16) | ti);
    }
    __syncthreads();

    // Loop-invariant operands that would otherwise be re-materialised from SGPRs inside every pair
    // iteration (a VOP3 instruction takes one scalar source): keep them in VGPRs.
    LJModel<real> mdl = a.model;
    LJUni<real> uni = a.uni;
    if (FAST) asm volatile("" : "+v"(uni.nx0), "+v"(uni.idl2), "+v"(uni.p4), "+v"(uni.p3), "+v"(uni.p0));
    else if (BITMASK == EMDEE_FORCES) asm volatile("" : "+v"(mdl.nx0), "+v"(mdl.idl2), "+v"(mdl.h4), "+v"(mdl.h3), "+v"(mdl.k6));
    else asm volatile("" : "+v"(mdl.x0), "+v"(mdl.k3));
    // SOA kernels index the coordinate planes with the BYTE offsets stored in the list; the others decode the slot
    const unsigned char *plane_b = reinterpret_cast<const unsigned char *>(s_dyn);
    constexpr int PLANE_BYTES = PITCH * (int)sizeof(real);

    // ---- own atoms: one G-lane group per atom; the NEXT atom's indices are fetched meanwhile -----
    const int gl = lane & (G - 1);                            // lane inside the group
    const int gid = (tid / WAVE) * (WAVE / G) + lane / G;     // group inside the block
    unsigned long long st_entries = 0, st_inside = 0;
    int st_max = 0;
    // The first NPF blocks of a row are prefetched one atom ahead (G = 4: 64 of the ~72 entries of an LJ row; the
    // block after them is requested when the atom's turn starts and arrives while those 64 are being worked on).
    constexpr int NPF = brick_prefetch_blocks(G, THREADS);   // (G = 4: 64 entries ahead, the rest of the row one block ahead)
    struct IdxBuf { uint4 q[NPF]; };
    // (a group past the brick's last atom reads the last atom's row: valid memory, no zero-fill and no branch; its own
    // position is the sentinel record, so nothing it reads passes the cutoff test, and nothing it computes is stored)
    auto fetch = [&](int o) {
        IdxBuf b;
        const unsigned short *row = a.nbr + (size_t)T.oinfo[min(o, n_own - 1)].x * a.stride + gl * EPL;
#pragma unroll
        for (int k = 0; k < NPF; k++) b.q[k] = *reinterpret_cast<const uint4 *>(row + k * BLK);   // stride >= (NPF + 1) BLK (host)
        return b;
    };
    IdxBuf nxt = fetch(gid);
    for (int ob = 0; ob < n_own; ob += NGROUPS) {             // wave-uniform trip count
        const int o = ob + gid;
        const bool have = o < n_own;
        const int2 info = have ? T.oinfo[o] : make_int2(0, 0);
        const int p = info.x, ti = info.y & 0xffff, m = (int)((unsigned)info.y >> 16);
        const IdxBuf cur = nxt;
        // rows longer than the prefetched blocks (rc = 3.5 sigma: 184 entries): the next block is requested
        // now and arrives while the first NPF blocks are being worked on
        // (rows are sentinel-padded up to the stride, and the stride holds at least NPF + 1 blocks: read whatever the row length)
        uint4 more = *reinterpret_cast<const uint4 *>(a.nbr + (size_t)p * a.stride + NPF * BLK + gl * EPL);
        nxt = fetch(o + NGROUPS);
        const int wm = wave_group_max<G>(m);
        real xi, yi, zi, hs_i, te_i;
        if (SOA) { xi = plane[ti]; yi = plane[PITCH + ti]; zi = plane[2 * PITCH + ti]; hs_i = te_i = 0; }
        else tile_load<real>(tile, tile_te, ti, xi, yi, zi, hs_i, te_i);
        real fx = 0, fy = 0, fz = 0, e = 0, w = 0;
        real vx = 0, vy = 0, vz = 0, bx = 0, by = 0, bz = 0, imv = 1, nx = 0, ny = 0, nz = 0;
        // (the SOA kernels fetch these after the pair loop instead: 14 registers less = three workgroups per CU)
        if (MODE == BRICK_STEP && !SOA && have && gl == G - 1) {   // owner lane: its loads fly during the pair loop
            vx = a.vel[p]; vy = a.vel[a.pitch + p]; vz = a.vel[2 * a.pitch + p];
            bx = a.xb[p]; by = a.xb[a.pitch + p]; bz = a.xb[2 * a.pitch + p];
            if (a.inv_mass) imv = a.inv_mass[p];
            if (a.noise) { nx = a.noise[p]; ny = a.noise[a.pitch + p]; nz = a.noise[2 * a.pitch + p]; }
        }
        // one block of 8 G neighbours: lane gl holds entries b0 + gl + t G, t = 0..7, in q
        // fp32 single-species kernels: two neighbours per lane and pass in packed arithmetic (v_pk_*_f32: the only
        // way past one fp32 operation per lane and instruction); the coordinate planes deliver (x_a, x_b) pairs
        // straight into adjacent registers
        f32x2 pfx = {0.f, 0.f}, pfy = {0.f, 0.f}, pfz = {0.f, 0.f}, pe = {0.f, 0.f}, pw = {0.f, 0.f};
        constexpr bool PACKED = SOA && sizeof(real) == 4;
        auto block2 = [&](const uint4 &q, int b0) {
            if constexpr (PACKED) {
#pragma unroll
                for (int t = 0; t < EPL; t += 2) {
                    if (b0 + t * G >= wm) break;              // wave-uniform; the partner entry t + 1 is a sentinel at worst
                    const unsigned char *pa = plane_b + pick16(q, t), *pb = plane_b + pick16(q, t + 1);   // byte offsets
                    auto ld = [](const unsigned char *p, int plane_no) { return *reinterpret_cast<const float *>(p + plane_no * PLANE_BYTES); };
                    const f32x2 dx = f32x2{(float)xi, (float)xi} - f32x2{ld(pa, 0), ld(pb, 0)};
                    const f32x2 dy = f32x2{(float)yi, (float)yi} - f32x2{ld(pa, 1), ld(pb, 1)};
                    const f32x2 dz = f32x2{(float)zi, (float)zi} - f32x2{ld(pa, 2), ld(pb, 2)};
                    const f32x2 r2 = dx * dx + dy * dy + dz * dz;
                    if constexpr (FAST) {                     // scaled coordinates, constants folded (LJUni)
                        const bool ina = r2.x < (float)uni.rc2, inb = r2.y < (float)uni.rc2;
                        if (ina | inb) {
                            f32x2 wr2 = lj_force_over_r2_uni2(r2, uni);
                            wr2 = f32x2{ina ? wr2.x : 0.f, inb ? wr2.y : 0.f};
                            pfx += wr2 * dx; pfy += wr2 * dy; pfz += wr2 * dz;
                        }
                        continue;
                    }
                    const bool ina = r2.x < (float)a.model.rc2, inb = r2.y < (float)a.model.rc2;   // strict test (Q2)
                    if (ina | inb) {
                        const f32x2 inv = {fast_rcp(r2.x), fast_rcp(r2.y)};
                        const f32x2 sg2 = {(float)a.uni_sigma2, (float)a.uni_sigma2}, e4v = {(float)a.uni_e4, (float)a.uni_e4};
                        if (BITMASK == EMDEE_FORCES) {
                            f32x2 wr2 = lj_force_over_r2_2(r2, inv, mdl, sg2, e4v);
                            wr2 = f32x2{ina ? wr2.x : 0.f, inb ? wr2.y : 0.f};
                            pfx += wr2 * dx; pfy += wr2 * dy; pfz += wr2 * dz;
                        } else {
                            f32x2 E, W;
                            lj_interaction_pair2(r2, inv, mdl, sg2, e4v, E, W);
                            E = f32x2{ina ? E.x : 0.f, inb ? E.y : 0.f};
                            W = f32x2{ina ? W.x : 0.f, inb ? W.y : 0.f};
                            if (BITMASK & EMDEE_FORCES) {
                                const f32x2 wr2 = W * inv;
                                pfx += wr2 * dx; pfy += wr2 * dy; pfz += wr2 * dz;
                            }
                            if (BITMASK & EMDEE_ENERGIES) pe += E;
                            if (BITMASK & EMDEE_VIRIALS) pw += W;
                        }
                    }
                }
            }
        };
        auto block = [&](const uint4 &q, int b0) {
            if constexpr (PACKED) {
                block2(q, b0);
                return;
            }
#ifndef EMDEE_PREFETCH_XJ
#define EMDEE_PREFETCH_XJ 1
#endif
            // (EMDEE_PREFETCH_XJ, on: the coordinates of entry t + 1 are requested before the arithmetic of entry t, as in the typed
            // kernels -- round 5, same-box A/B of the build: fused launch 1.290 -> 1.280 ms, 600 -> 603.6 steps/s; =0 is the A/B baseline)
            // Force-only launches (the integrator's): with energies or virials as well the six registers of the read-ahead take the fp64
            // kernel from 76 to 82 VGPRs and a third workgroup off the CU -- the operator's F+E+W call went 1.49 -> 1.53 ms for it
            // (found by running round 4's tree beside this one, profiles/r05/r04_vs_r05_same_box.txt)
            constexpr bool AHEAD = SOA && EMDEE_PREFETCH_XJ != 0 && BITMASK == EMDEE_FORCES;
            real xn = 0, yn = 0, zn = 0;
            if (AHEAD) {
                const unsigned char *pn = plane_b + pick16(q, 0);
                xn = *reinterpret_cast<const real *>(pn); yn = *reinterpret_cast<const real *>(pn + PLANE_BYTES);
                zn = *reinterpret_cast<const real *>(pn + 2 * PLANE_BYTES);
            }
#pragma unroll
            for (int t = 0; t < EPL; t++) {
                if (b0 + t * G >= wm) break;                  // wave-uniform; entries past a row's end are sentinels
                {
                    const int sj = SOA ? pick16(q, t) : (pick16(q, t) >> a.idx_shift);
                    real xj, yj, zj, hs_j, te_j;
                    if (AHEAD) {
                        xj = xn; yj = yn; zj = zn;
                        if (t + 1 < EPL) {
                            const unsigned char *pj = plane_b + pick16(q, t + 1);
                            xn = *reinterpret_cast<const real *>(pj); yn = *reinterpret_cast<const real *>(pj + PLANE_BYTES);
                            zn = *reinterpret_cast<const real *>(pj + 2 * PLANE_BYTES);
                        }
                        hs_j = te_j = 0;
                    } else if (SOA) {
                        const unsigned char *pj = plane_b + sj;          // byte offset: three reads off one address register
                        xj = *reinterpret_cast<const real *>(pj);
                        yj = *reinterpret_cast<const real *>(pj + PLANE_BYTES);
                        zj = *reinterpret_cast<const real *>(pj + 2 * PLANE_BYTES);
                        hs_j = te_j = 0;
                    } else tile_load<real>(tile, tile_te, sj, xj, yj, zj, hs_j, te_j);
                    real dx = xi - xj, dy = yi - yj, dz = zi - zj;
                    if (REL && MODE == BRICK_FORCE && !SOA) {
                        if (refmath) {   // r_ij = L minimum_image(s_i - s_j), src/nonbonded.jl:40,70
                            dx = a.g.len[0] * (dx - rint(dx)); dy = a.g.len[1] * (dy - rint(dy)); dz = a.g.len[2] * (dz - rint(dz));
                        }
                    }
                    const real r2 = dx * dx + dy * dy + dz * dz;
                    if (MODE == BRICK_STATS) {
                        st_inside += (b0 + t * G + gl < m && r2 < a.model.rc2) ? 1ull : 0ull;
                    } else if (FAST) {
                        if (r2 < uni.rc2) {                   // strict test (Q2), in scaled coordinates
                            const real wr2 = lj_force_over_r2_uni(r2, uni);
                            fx += wr2 * dx; fy += wr2 * dy; fz += wr2 * dz;
                        }
                    } else if (r2 < a.model.rc2) {            // strict test (Q2)
                        const real inv_r2 = fast_rcp(r2);
                        if (BITMASK == EMDEE_FORCES) {        // the MD loop's kernels: W / r2 directly (lj_pair.hpp)
                            real wr2;
                            if (UNI) {
                                wr2 = lj_force_over_r2(r2, inv_r2, mdl, a.uni_sigma2, a.uni_e4);
                            } else {
                                const real sg = hs_i + hs_j;
                                wr2 = lj_force_over_r2(r2, inv_r2, mdl, sg * sg, te_i * te_j);
                            }
                            fx += wr2 * dx; fy += wr2 * dy; fz += wr2 * dz;
                        } else {
                            real E, W;
                            if (UNI) {   // single species: no parameter gather, conversion or mixing per pair
                                lj_interaction_pair(r2, inv_r2, mdl, a.uni_sigma2, a.uni_e4, E, W);
                            } else {
                                lj_interaction(r2, inv_r2, mdl, hs_i, te_i, hs_j, te_j, E, W);
                            }
                            if (BITMASK & EMDEE_FORCES) {
                                const real wr2 = W * inv_r2;      // src/nonbonded.jl:139
                                fx += wr2 * dx; fy += wr2 * dy; fz += wr2 * dz;
                            }
                            if (BITMASK & EMDEE_ENERGIES) e += E;
                            if (BITMASK & EMDEE_VIRIALS) w += W;
                        }
                    }
                }
            }
        };
#pragma unroll
        for (int k = 0; k < NPF; k++)
            if (k * BLK < wm) block(cur.q[k], k * BLK);
        for (int b0 = NPF * BLK; b0 < wm; b0 += BLK) {        // longer rows: one block ahead
            const uint4 q = more;
            more = make_uint4(0, 0, 0, 0);
            if (b0 + BLK < a.stride) more = *reinterpret_cast<const uint4 *>(a.nbr + (size_t)p * a.stride + b0 + BLK + gl * EPL);
            block(q, b0);
        }
        if (PACKED) {
            fx = (real)(pfx.x + pfx.y); fy = (real)(pfy.x + pfy.y); fz = (real)(pfz.x + pfz.y);
            e = (real)(pe.x + pe.y); w = (real)(pw.x + pw.y);
        }
        if (MODE == BRICK_STATS) {
            if (gl == 0) { st_entries += (unsigned long long)m; st_max = max(st_max, m); }
        } else {
            // all lanes are active here: DPP reductions see every lane of the group
            if (BITMASK & EMDEE_FORCES) {
                fx = group_sum_to_last<G>(fx); fy = group_sum_to_last<G>(fy); fz = group_sum_to_last<G>(fz);
                if (FAST) { fx *= uni.inv_sigma; fy *= uni.inv_sigma; fz *= uni.inv_sigma; }   // back from scaled coordinates
            }
            if (BITMASK & EMDEE_ENERGIES) e = group_sum_to_last<G>(e);
            if (BITMASK & EMDEE_VIRIALS) w = group_sum_to_last<G>(w);
            if (MODE == BRICK_STEP) {
                if (have && gl == G - 1) {
                    if (SOA) {
                        vx = a.vel[p]; vy = a.vel[a.pitch + p]; vz = a.vel[2 * a.pitch + p];
                        bx = a.xb[p]; by = a.xb[a.pitch + p]; bz = a.xb[2 * a.pitch + p];
                        if (a.inv_mass) imv = a.inv_mass[p];
                        if (a.noise) { nx = a.noise[p]; ny = a.noise[a.pitch + p]; nz = a.noise[2 * a.pitch + p]; }
                    }
                    const real cm = a.kick_c * imv;
                    vx += cm * fx; vy += cm * fy; vz += cm * fz;
                    // Langevin O step; NVE launches pass c1 = 1 and the noise registers stay 0: v 1 + 0 is v, bit for bit,
                    // and the three fmas cost less than the selects an `if (a.noise)` is compiled into
                    vx = a.lgv_c1 * vx + nx; vy = a.lgv_c1 * vy + ny; vz = a.lgv_c1 * vz + nz;
                    a.vel_next[p] = vx; a.vel_next[a.pitch + p] = vy; a.vel_next[2 * a.pitch + p] = vz;
                    Rec<real> r = a.rec[p];                    // keeps the LJAtom fields bit for bit
                    // (the tile holds scaled coordinates in the FAST kernels and brick-relative ones in fp32 boxes)
                    if (FAST || REL) { r.x += a.dt * vx; r.y += a.dt * vy; r.z += a.dt * vz; }
                    else { r.x = xi + a.dt * vx; r.y = yi + a.dt * vy; r.z = zi + a.dt * vz; }
                    a.rec_next[p] = r;
                    const real ex = r.x - bx, ey = r.y - by, ez = r.z - bz;
                    if (ex * ex + ey * ey + ez * ez > a.thr2) *a.trigger = 1;
                    if (far_skip && ex * ex + ey * ey + ez * ez > a.thr2_near) *a.far_word = 1;
                }
            } else if (have && gl == G - 1) {
                if (a.user_f != nullptr || a.user_e != nullptr || a.user_w != nullptr) {
                    const size_t i = (size_t)a.perm[p];                     // caller index of this atom
                    // (a tuning variant may run the all-outputs kernel for a narrower request: unselected arrays are NULL)
                    if ((BITMASK & EMDEE_FORCES) && a.user_f) { a.user_f[3 * i] = fx; a.user_f[3 * i + 1] = fy; a.user_f[3 * i + 2] = fz; }
                    if ((BITMASK & EMDEE_ENERGIES) && a.user_e) a.user_e[i] = (real)0.5 * e;
                    if ((BITMASK & EMDEE_VIRIALS) && a.user_w) a.user_w[i] = (real)0.5 * w;
                } else {
                    if (BITMASK & EMDEE_FORCES) {
                        a.frc[p] = fx; a.frc[a.pitch + p] = fy; a.frc[2 * a.pitch + p] = fz;
                    }
                    if (BITMASK & EMDEE_ENERGIES) a.en[p] = (real)0.5 * e;   // src/nonbonded.jl:142-145
                    if (BITMASK & EMDEE_VIRIALS) a.vir[p] = (real)0.5 * w;
                }
            }
        }
    }
    if (MODE == BRICK_STATS) {
        // exact integer totals; reduced over the wavefront first (three hot addresses for the whole grid)
#pragma unroll
        for (int off = 1; off < WAVE; off <<= 1) {
            st_entries += __shfl_xor(st_entries, off);
            st_inside += __shfl_xor(st_inside, off);
            st_max = max(st_max, __shfl_xor(st_max, off));
        }
        // ... then over the workgroup in LDS (the tile is no longer needed): same-address atomics serialise in L2 at
        // ~10 ns each, one per wavefront was still 9 of this kernel's 10.6 ms at 10^7 atoms
        __syncthreads();
        unsigned long long *red = reinterpret_cast<unsigned long long *>(s_dyn);
        if (lane == 0) { red[3 * (tid / WAVE)] = st_entries; red[3 * (tid / WAVE) + 1] = (unsigned long long)st_max; red[3 * (tid / WAVE) + 2] = st_inside; }
        __syncthreads();
        if (tid == 0) {
            unsigned long long se = 0, sm = 0, si = 0;
            for (int wv = 0; wv < THREADS / WAVE; wv++) { se += red[3 * wv]; sm = max(sm, red[3 * wv + 1]); si += red[3 * wv + 2]; }
            if (se) atomicAdd(&a.stats[0], se);
            if (sm) atomicMax(&a.stats[1], sm);
            if (si) atomicAdd(&a.stats[2], si);
        }
    }
}

// Debug/verification accessor: the neighbour rows as CALLER ids (tile-local slots decoded through the brick's tables),
// counts[i] entries at out[i * capacity ...], in list order.  One workgroup per brick, like the force kernels.
template <typename real, class Shape, int THREADS, int G>
__global__ __launch_bounds__(THREADS) void k_brick_export(BrickArgs<real> a, int *__restrict__ counts, int *__restrict__ out,
                                                          int capacity, const int *__restrict__ cmap = nullptr) {
    extern __shared__ __attribute__((aligned(16))) unsigned char s_dyn[];
    BrickTables<Shape, THREADS> T;
    T.carve(s_dyn);
    int bxi, byi, bzi, tile_n, n_own;
    if (!brick_setup<real, Shape, THREADS>(a, T, bxi, byi, bzi, tile_n, n_own)) return;
    for (int o = threadIdx.x; o < n_own; o += THREADS) {
        int ti, p;
        brick_locate(T, o, ti, p);
        if (a.perm[p] >= a.n_owned) continue;
        const int i = cmap ? cmap[a.perm[p]] : a.perm[p];
        const int m = a.far_skip ? (a.cnt[p] & 255) : a.cnt[p];
        counts[i] = m;
        const unsigned short *row = a.nbr + (size_t)p * a.stride;
        for (int e = 0; e < m && e < capacity; e++) {
            const int sl = (int)row[row_position<G>((unsigned)e)] >> a.idx_shift;
            int lo = 0, hi = Shape::NTC;                  // tile cell with off[tc] <= sl < off[tc + 1]
            while (hi - lo > 1) {
                const int mid = (lo + hi) >> 1;
                if (T.off[mid] <= sl) lo = mid; else hi = mid;
            }
            const int j = a.perm[T.gbeg[lo] + (sl - T.off[lo])];
            out[(size_t)i * capacity + e] = cmap ? cmap[j] : j;
        }
    }
}

// Rows without their excluded entries (kernels.hpp, "exclusions and 1-4 pairs"): run right after a build when the caller
// named pairs to leave out.  One thread per own atom that has exclusions: its entries are decoded as k_brick_export does
// (tile slot -> cell-order slot -> caller id), looked up in the atom's sorted exclusion list, and the row is compacted in
// place, the vacated tail refilled with the sentinel.  Untyped rows only (a box with exclusions keeps the general kernels).
template <typename real, class Shape, int THREADS, int G>
__global__ __launch_bounds__(THREADS) void k_brick_filter(BrickArgs<real> a, const int *__restrict__ ex_start,
                                                          const int *__restrict__ ex_idx) {
    extern __shared__ __attribute__((aligned(16))) unsigned char s_dyn[];
    BrickTables<Shape, THREADS> T;
    T.carve(s_dyn);
    int bxi, byi, bzi, tile_n, n_own;
    if (!brick_setup<real, Shape, THREADS>(a, T, bxi, byi, bzi, tile_n, n_own)) return;
    for (int o = threadIdx.x; o < n_own; o += THREADS) {
        int ti, p;
        brick_locate(T, o, ti, p);
        const int i = a.perm[p];
        if (i >= a.n_owned) continue;
        const int lo_x = ex_start[i], hi_x = ex_start[i + 1];
        if (lo_x == hi_x) continue;
        const int m = min(a.cnt[p], a.stride);
        unsigned short *row = a.nbr + (size_t)p * a.stride;
        int w = 0;
        for (int e = 0; e < m; e++) {
            const unsigned short ent = row[row_position<G>((unsigned)e)];
            const int sl = (int)ent >> a.idx_shift;
            int lo = 0, hi = Shape::NTC;                  // tile cell with off[tc] <= sl < off[tc + 1]
            while (hi - lo > 1) {
                const int mid = (lo + hi) >> 1;
                if (T.off[mid] <= sl) lo = mid; else hi = mid;
            }
            const int j = a.perm[T.gbeg[lo] + (sl - T.off[lo])];
            if (!csr_holds(ex_idx, lo_x, hi_x, j)) {
                if (w != e) row[row_position<G>((unsigned)w)] = ent;
                w++;
            }
        }
        for (int e = w; e < m; e++) row[row_position<G>((unsigned)e)] = 0;   // the sentinel slot
        a.cnt[p] = w;
    }
}

// Largest tile (brick + halo population) and largest own population over all bricks -> sizes the
// dynamic LDS of the brick kernels.  out[0] = max tile, out[1] = max own, out[2] = most atoms in three
// consecutive cells of a tile row (what one group of the build kernel scans per row).
template <class Shape>
__global__ void k_brick_tile_max(BrickGrid bg, int Mx, int My, int Mz, int px, int py, int pz,
                                 const int *__restrict__ start, int *__restrict__ out) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    int total = 0, own = 0, span3 = 0;
    if (b < bg.nbricks) {
        const int bxi = b % bg.nb[0], byi = (b / bg.nb[0]) % bg.nb[1], bzi = b / (bg.nb[0] * bg.nb[1]);
        const int ox0 = bxi * Shape::BX, oy0 = byi * Shape::BY, oz0 = bzi * Shape::BZ;
        const int ox1 = min(ox0 + Shape::BX, Mx), oy1 = min(oy0 + Shape::BY, My), oz1 = min(oz0 + Shape::BZ, Mz);
        for (int tz = 0; tz < Shape::TZ; tz++)
            for (int ty = 0; ty < Shape::TY; ty++) {
                const int ry = oy0 - 1 + ty, rz = oz0 - 1 + tz;
                int gy = ry, gz = rz;
                if (gy > oy1 || gz > oz1) continue;
                if (gy < 0) { if (!py) continue; gy += My; } else if (gy >= My) { if (!py) continue; gy -= My; }
                if (gz < 0) { if (!pz) continue; gz += Mz; } else if (gz >= Mz) { if (!pz) continue; gz -= Mz; }
                int p1 = 0, p2 = 0;   // the two previous cells of this tile row
                for (int tx = 0; tx < Shape::TX; tx++) {
                    const int rx = ox0 - 1 + tx;
                    int gx = rx;
                    bool valid = gx <= ox1;
                    if (gx < 0) { if (!px) valid = false; gx += Mx; } else if (gx >= Mx) { if (!px) valid = false; gx -= Mx; }
                    int pop = 0;
                    if (valid) {
                        const int c = gx + Mx * (gy + My * gz);
                        pop = start[c + 1] - start[c];
                        total += pop;
                        if (rx >= ox0 && rx < ox1 && ry >= oy0 && ry < oy1 && rz >= oz0 && rz < oz1) own += pop;
                    }
                    span3 = max(span3, pop + p1 + p2);
                    p2 = p1; p1 = pop;
                }
            }
    }
#pragma unroll
    for (int off = 1; off < WAVE; off <<= 1) {
        total = max(total, __shfl_xor(total, off));
        own = max(own, __shfl_xor(own, off));
        span3 = max(span3, __shfl_xor(span3, off));
    }
    if ((threadIdx.x & (WAVE - 1)) == 0 && total > 0) {
        atomicMax(&out[0], total);
        atomicMax(&out[1], own);
        atomicMax(&out[2], span3);
    }
}

}  // namespace emdee
