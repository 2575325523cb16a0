// kernels.hpp -- device kernels of the O(N) nonbonded path, templated on the real type.
//
// Data layout in HBM (cell order = atoms sorted by cell id, ascending caller id inside a cell):
//   rec[p]      one aligned record per atom: fp64 {x,y,z, half_sigma, twice_sqrt_eps} = 32 B
//               (a neighbour gather is exactly one record: position AND LJAtom parameters);
//               fp32 {x,y,z, half_sigma} = 16 B plus te[p] (4 B)
//   vel/frc     structure-of-arrays, 3 planes of `pitch` reals: every streaming pass is a
//               unit-stride 4/8-B-per-lane access
//   nbr         row-major ELL: row p holds the neighbours of atom p at nbr[p*stride .. +cnt[p])
//               so a wavefront reads its atom's indices as 256-B coalesced segments
//   perm/inv    cell-order slot -> caller id and back
//
// Reference lines restated here: cell id convention src/cells.jl:82-85,180-181; minimum image
// src/nonbonded.jl:40; f_ij = W/r2 * r_ij and the half split of E and W src/nonbonded.jl:136-145;
// the pair function is lj_pair.hpp.
#pragma once

#include <hip/hip_runtime.h>

#include "common.hpp"
#include "lj_pair.hpp"
#include "wave_ops.hpp"

namespace emdee {

// ------------------------------------------------------------------------------------ records
template <typename real>
struct Rec;
template <>
struct alignas(32) Rec<double> {
    double x, y, z;
    float hs, te;
};
template <>
struct alignas(16) Rec<float> {
    float x, y, z, hs;
};

// Cell-relative records (round 5; fp32 integrators): a record holds the atom's position RELATIVE TO THE ORIGIN OF ITS CELL
// as of the last sort, so that the drift x += dt v of a step rounds at the ulp of a cell-sized number (2.4e-7 sigma) and
// not at the ulp of the box (1.5e-5 sigma at 10^7 atoms, 3.1e-5 at 10^8 -- against ~5e-3 sigma moved per step: the noise
// that random-walked the fp32 runs' energy).  Between sorts an atom may wander a little out of its cell: the relative
// coordinate simply leaves [0, cell width).  on = 0: records are absolute (every fp64 state, operator handles, domains).
struct RelGrid {
    double lo[3], cw[3];      // box origin, cell widths
    int M[3];
    int on;
    const int *cell;          // cell of every slot (NbSystem::cell_sorted as of the sort the records belong to)
    // Cell origins sit on ONE fixed-point grid of the box, REL_FX = 2^19 points per length unit (what fp32 resolves at the far end
    // of a brick's tile): a tile coordinate is then (round(record x 2^19) + an INTEGER per tile cell) / 2^19, exact in fp32, and
    // two atoms have the same separation in whichever brick's tile they meet -- F_ij = -F_ji to the bit (brick.hpp rel_tile).
    __host__ __device__ __forceinline__ double origin_q(int d, int c) const { return rint((lo[d] + (double)c * cw[d]) * 524288.0); }   // in grid points
    // c / m and c % m for 0 <= c < 2^24 (a box of 10^8 atoms has 5.6e6 cells): one fp32 multiply and a correction by one instead of
    // the ~35 instructions of an integer division by a run-time divisor -- twice per atom in every pass that reads relative records
    __device__ static __forceinline__ int divmod(int c, int m, int &rem) {
        int q = (int)((float)c * __builtin_amdgcn_rcpf((float)m));       // (v_rcp_f32: 1 ulp, the correction below absorbs it)
        rem = c - q * m;
        while (rem < 0) { q--; rem += m; }                   // (the estimate is off by at most one: c < 2^24, two roundings of 6e-8)
        while (rem >= m) { q++; rem -= m; }
        return q;
    }
    __device__ __forceinline__ void origin(int c, double &ox, double &oy, double &oz) const {
        int cx, cy, cz;
        if (c < (1 << 24)) {
            const int t = divmod(c, M[0], cx);
            cz = divmod(t, M[1], cy);
        } else {
            cx = c % M[0]; cy = (c / M[0]) % M[1]; cz = c / (M[0] * M[1]);
        }
        ox = origin_q(0, cx) * (1.0 / 524288.0); oy = origin_q(1, cy) * (1.0 / 524288.0); oz = origin_q(2, cz) * (1.0 / 524288.0);
    }
};
constexpr float REL_FX = 524288.f, REL_IFX = 1.f / 524288.f;

// read-only view of the cell-ordered atoms
template <typename real>
struct AtomView {
    const Rec<real> *rec;
    const float *te;   // fp32 only (fp64 keeps te inside the record)
    RelGrid rel;       // fp32 only: cell-relative records (load_atom returns absolute coordinates, rounded to fp32)
};

__device__ __forceinline__ void load_atom(const AtomView<double> &a, int p, double &x, double &y, double &z, double &hs,
                                          double &te) {
    Rec<double> r = a.rec[p];   // 2 x global_load_dwordx4
    x = r.x; y = r.y; z = r.z; hs = (double)r.hs; te = (double)r.te;
}
__device__ __forceinline__ void load_atom(const AtomView<float> &a, int p, float &x, float &y, float &z, float &hs,
                                          float &te) {
    Rec<float> r = a.rec[p];    // 1 x global_load_dwordx4
    x = r.x; y = r.y; z = r.z; hs = r.hs; te = a.te[p];
    if (a.rel.on) {             // (the kernels that come through here are not the integrator's hot ones)
        double ox, oy, oz;
        a.rel.origin(a.rel.cell[p], ox, oy, oz);
        x = (float)((double)x + ox); y = (float)((double)y + oy); z = (float)((double)z + oz);
    }
}

// ------------------------------------------------------------------------------------ grid
template <typename real>
struct GridP {
    real lo[3], len[3];      // box
    real plen[3], pinv[3];   // minimum image d -= plen * rint(d * pinv); both 0 on non-periodic dims
    int per[3];
    int M[3];                // cells per dimension
    int nd;                  // stencil half-width in cells (cell side >= rlist / nd)
    int one_based;           // Cells API: ids are 1-based (src/cells.jl:181)
};

// 0-based voxel of coordinate p along dimension d: floor(M (s - floor s)), s = (p - lo)/len.
// True division, not a reciprocal multiply: the Cells API must agree bit-for-bit with the
// reference arithmetic s = r / L (src/cells.jl:79-84,180).
template <typename real>
__device__ __forceinline__ int voxel(real p, real lo, real len, int M, int periodic) {
    real s = (p - lo) / len;
    real t = periodic ? s - floor(s) : s;
    int v = (int)floor((real)M * t);
    return min(max(v, 0), M - 1);
}

template <typename real>
__device__ __forceinline__ int cell_id(const GridP<real> &g, real x, real y, real z) {
    int vx = voxel(x, g.lo[0], g.len[0], g.M[0], g.per[0]);
    int vy = voxel(y, g.lo[1], g.len[1], g.M[1], g.per[1]);
    int vz = voxel(z, g.lo[2], g.len[2], g.M[2], g.per[2]);
    return vx + g.M[0] * (vy + g.M[1] * vz);
}

template <typename real>
__device__ __forceinline__ real min_image(real d, real plen, real pinv) {
    return d - plen * rint(d * pinv);   // v_rndne: ties-to-even like Julia round (Q8)
}

// position sources: caller order (3xN interleaved) or cell-ordered records
template <typename real>
struct UserPos {
    const real *p;
    __device__ __forceinline__ void get(int i, real &x, real &y, real &z) const {
        x = p[3 * (size_t)i]; y = p[3 * (size_t)i + 1]; z = p[3 * (size_t)i + 2];
    }
};
template <typename real>
struct RecPos {
    const Rec<real> *r;
    RelGrid rel;
    __device__ __forceinline__ void get(int i, real &x, real &y, real &z) const {
        Rec<real> q = r[i];
        x = q.x; y = q.y; z = q.z;
        if (sizeof(real) == 4 && rel.on) {
            double ox, oy, oz;
            rel.origin(rel.cell[i], ox, oy, oz);
            x = (real)((double)q.x + ox); y = (real)((double)q.y + oy); z = (real)((double)q.z + oz);
        }
    }
};

// ------------------------------------------------------------------------------------ x sub-bins (untyped boxes)
// Inside a cell the atoms are ordered by the quarter of the cell (along x) they sit in: the sort digit is cell * S + sub-bin,
// exactly as a typed box uses cell * 2 + species.  The tile rows of the neighbour build are runs of three cells along x, so
// an atom in sub-bin s needs only the sub-bins >= s + K of the cell on its left and <= s - K of the cell on its right
// (K = 0 when the cell is as wide as r_list): 9 of the 12 quarters of a row instead of all -- the build tests a quarter
// fewer candidates (brick.hpp).  The sub-bin comes from the SAME product M t the cell does (voxel above), so an atom of
// sub-bin s of cell v has M t in [v + s / S, v + (s + 1) / S) exactly.
template <typename real, class Src>
struct XSubBin {
    Src src;
    real lo, len;
    int M, per, S;
    __device__ __forceinline__ int of(int i) const {
        real x, y, z;
        src.get(i, x, y, z);
        const real s = (x - lo) / len;
        const real t = per ? s - floor(s) : s;
        const real f = (real)M * t;
        const int v = (int)floor(f);
        if (v < 0) return 0;                       // (clamped into the first / last cell by voxel())
        if (v > M - 1) return S - 1;
        const int sb = (int)((f - (real)v) * (real)S);
        return min(max(sb, 0), S - 1);
    }
};

// ------------------------------------------------------------------------------------ species ("typed" boxes)
// A box with 2 distinct LJAtom values is sorted by (cell, species): inside a cell the atoms of species 0 come first.
// The LDS tiles of the brick kernels are then staged species-major, neighbour rows come out as one segment per neighbour
// species, and the pair loop needs no per-pair parameter mixing (brick.hpp, NT).  The table holds the distinct LJAtom
// bit patterns in ascending order (the host sorts what k_species_collect found: the numbering must not depend on a race).
constexpr int MAX_SPECIES = 4;
struct SpeciesTable {
    int n;                                  // 1: untyped (every kernel behaves as before)
    unsigned long long key[MAX_SPECIES];    // half_sigma bits | twice_sqrt_eps bits << 32
};
__host__ __device__ __forceinline__ unsigned long long species_key(float hs, float te) {
    unsigned a, b;
#if defined(__HIP_DEVICE_COMPILE__)
    a = __float_as_uint(hs); b = __float_as_uint(te);
#else
    memcpy(&a, &hs, 4); memcpy(&b, &te, 4);
#endif
    return (unsigned long long)a | ((unsigned long long)b << 32);
}
__device__ __forceinline__ int species_of(const SpeciesTable &t, unsigned long long k) {
    int s = 0;
#pragma unroll
    for (int q = 1; q < MAX_SPECIES; q++) s = (q < t.n && t.key[q] == k) ? q : s;
    return s;
}
// species of item i of a position source (caller order: from the LJAtom array; cell order: from the records)
struct NoSpecies {
    __device__ __forceinline__ int of(int) const { return 0; }
};
struct UserSpecies {
    SpeciesTable t;
    const emdee_lj_atom *atoms;
    __device__ __forceinline__ int of(int i) const { return species_of(t, species_key(atoms[i].half_sigma, atoms[i].twice_sqrt_eps)); }
};
template <typename real>
struct RecSpecies;
template <>
struct RecSpecies<double> {
    SpeciesTable t;
    const Rec<double> *rec;
    const float *te;
    __device__ __forceinline__ int of(int i) const { return species_of(t, species_key(rec[i].hs, rec[i].te)); }
};
template <>
struct RecSpecies<float> {
    SpeciesTable t;
    const Rec<float> *rec;
    const float *te;
    __device__ __forceinline__ int of(int i) const { return species_of(t, species_key(rec[i].hs, te[i])); }
};
// two-species boxes whose build takes x sub-bins (typed.hpp): digit = species * S + quarter
template <class Spc, class Sub>
struct SpeciesSub {
    Spc spc;
    Sub sub;
    __device__ __forceinline__ int of(int i) const { return spc.of(i) * sub.S + sub.of(i); }
};
// distinct LJAtom values of a box, at most MAX_SPECIES: tab[0..3] = keys in order of arrival (EMPTY = all ones),
// tab[4] != 0 if there are more.  Almost every thread finds its key with device-scope loads; the atomics are for the
// first few arrivals.
// (Round 4: ONE look-up per distinct key of a wavefront, by the first lane that holds it -- every thread looking at the hot words
// itself was 10^7 device-scope loads of four addresses: 2-6 ms at every load of a 10^7-atom state, more than its whole sort.)
static __global__ void k_species_collect(int n, const emdee_lj_atom *__restrict__ atoms, unsigned long long *__restrict__ tab) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const unsigned long long EMPTY = ~0ull;
    unsigned long long mine = EMPTY;
    bool pending = i < n;
    if (pending) mine = species_key(atoms[i].half_sigma, atoms[i].twice_sqrt_eps);
    const int lane = lane_id();
    for (int round = 0; round <= MAX_SPECIES; round++) {       // at most MAX_SPECIES + 1 distinct keys matter: one more is "too many"
        const unsigned long long left = __ballot(pending);
        if (left == 0) return;
        const int leader = __ffsll((long long)left) - 1;
        const unsigned lo = __shfl((unsigned)mine, leader), hi = __shfl((unsigned)(mine >> 32), leader);
        const unsigned long long key = (unsigned long long)lo | ((unsigned long long)hi << 32);
        if (pending && mine == key) {
            pending = false;
            if (lane == leader) {
                bool found = false;
                for (int q = 0; q < MAX_SPECIES && !found; q++) {
                    unsigned long long cur = __hip_atomic_load(&tab[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (cur == EMPTY) cur = atomicCAS(&tab[q], EMPTY, key);
                    found = cur == EMPTY || cur == key;
                }
                if (!found) tab[MAX_SPECIES] = 1;
            }
        }
    }
    if (__ballot(pending) != 0 && lane == 0) tab[MAX_SPECIES] = 1;   // a wavefront with more distinct keys than the table holds
}
// ------------------------------------------------------------------------------------ binning
// Runs of equal keys among the 64 lanes of a wavefront (adjacent lanes only).  When the input is already
// nearly cell-ordered -- every MD rebuild -- a wavefront spans ~4 cells, so one atomic per RUN instead of one
// per atom cuts the integer atomics ~16x.  Returns the first lane and the length of this lane's run.
__device__ __forceinline__ void wave_run(int key, int &first, int &len) {
    const int lane = lane_id();
    const int prev = __shfl_up(key, 1);
    const unsigned long long leaders = __ballot(lane == 0 || key != prev);
    const unsigned long long upto = leaders & (~0ull >> (63 - lane));          // leaders at or below me
    first = 63 - __clzll(upto);
    const unsigned long long after = (first == 63) ? 0ull : (leaders & (~0ull << (first + 1)));
    len = (after ? (__ffsll((long long)after) - 1) : 64) - first;
}

// Radix-count pass of the counting sort: one digit = the cell id.
// (typed boxes: the digit is cell * nt + species, so that a cell's atoms come out grouped by species)
// (keep, optional: one byte per item -- an item whose byte is 0 is not part of the new state: a decomposed domain re-sorts
// its own records with the leavers and the old ghosts struck out and the arrivals appended, NbSystem::resort_edit)
template <typename real, class Src, class Spc = NoSpecies>
__global__ void k_cell_assign(int n, Src src, GridP<real> g, int *__restrict__ cell_of, int *__restrict__ count, Spc spc = Spc(),
                              int nt = 1, const unsigned char *__restrict__ keep = nullptr) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    int c = -1;                                   // lanes past the end form their own run and add nothing
    if (i < n) {
        if (keep == nullptr || keep[i] != 0) {
            real x, y, z;
            src.get(i, x, y, z);
            c = cell_id(g, x, y, z) * nt + spc.of(i);
            cell_of[i] = c + g.one_based;
        } else {
            cell_of[i] = -1;
        }
    }
    int first, len;
    wave_run(c, first, len);
    if (c >= 0 && lane_id() == first) atomicAdd(&count[c], len);
}

// Scatter ids into their cell's range (arrival order inside a cell is arbitrary here)...
static __global__ void k_cell_scatter(int n, const int *__restrict__ cell_of, int one_based, const int *__restrict__ start,
                               int *__restrict__ fill, int *__restrict__ tmp) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int c = (i < n) ? cell_of[i] - one_based : -1;
    int first, len;
    wave_run(c, first, len);
    int base = 0;
    if (c >= 0 && lane_id() == first) base = start[c] + atomicAdd(&fill[c], len);
    base = __shfl(base, first);
    if (c >= 0) tmp[base + (lane_id() - first)] = i;
}

// The same for the engine's rebuilds, with each id's sort key (caller id) stored next to it: the ranking pass below then
// reads its cell-mates' keys from consecutive addresses instead of one dependent gather per cell-mate (cells of side
// >= r_list hold ~18 atoms: 0.22 -> 0.1 ms per rebuild at 10^7 atoms).
// (tag, optional: 64-bit ids that travel with the atoms -- decomposed domains order a cell's atoms by the low word of the
// GLOBAL id, which makes the cell order independent of the local numbering, i.e. of the history of migrations)
static __global__ void k_cell_scatter_keyed(int n, const int *__restrict__ cell_of, const int *__restrict__ start,
                                            int *__restrict__ fill, const int *__restrict__ key, int2 *__restrict__ tmp,
                                            const long long *__restrict__ tag = nullptr) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int c = (i < n) ? cell_of[i] : -1;
    int first, len;
    wave_run(c, first, len);
    int base = 0;
    if (c >= 0 && lane_id() == first) base = start[c] + atomicAdd(&fill[c], len);
    base = __shfl(base, first);
    if (c >= 0) tmp[base + (lane_id() - first)] = make_int2(i, tag ? (int)tag[i] : (key ? key[i] : i));
}
// (n_dev, optional: the number of items that were scattered, as a device word -- a re-sort with struck-out items launches over
// the upper bound; equal keys -- only the low words of two 64-bit tags can collide -- are ranked by the item index)
// (ties: keys that are caller ids or slots of the previous order are unique and the loop reads the keys alone, as in rounds 1-4
// -- 0.06 ms per rebuild at 10^7 atoms against 0.095 with the tie-break)
static __global__ void k_cell_rankfix_keyed(int n, const int *__restrict__ cell_of, const int *__restrict__ start,
                                            const int2 *__restrict__ tmp, int *__restrict__ order,
                                            const int *__restrict__ n_dev = nullptr, int ties = 1) {
    int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (n_dev) n = min(n, *n_dev);
    if (q >= n) return;
    const int2 me = tmp[q];
    const int c = cell_of[me.x];
    const int s = start[c], e = start[c + 1];
    int rank = 0;
    if (ties) {
        for (int r = s; r < e; r++) {
            const int2 o = tmp[r];
            rank += (o.y < me.y || (o.y == me.y && o.x < me.x)) ? 1 : 0;
        }
    } else {
        for (int r = s; r < e; r++) rank += (tmp[r].y < me.y) ? 1 : 0;
    }
    order[s + rank] = me.x;
}

// ...then make it deterministic: inside each cell order by key (caller id), by counting smaller
// keys among the cell-mates (cells hold O(10) atoms).  order[q] = source index of slot q.
static __global__ void k_cell_rankfix(int n, const int *__restrict__ cell_of, int one_based, const int *__restrict__ start,
                               const int *__restrict__ tmp, const int *__restrict__ key, int *__restrict__ order) {
    int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= n) return;
    int i = tmp[q];
    int c = cell_of[i] - one_based;
    int s = start[c], e = start[c + 1];
    int ki = key ? key[i] : i;
    int rank = 0;
    for (int r = s; r < e; r++) {
        int j = tmp[r];
        int kj = key ? key[j] : j;
        rank += (kj < ki) ? 1 : 0;
    }
    order[s + rank] = i;
}

// ------------------------------------------------------------------------------------ scan
constexpr int SCAN_THREADS = 256;
constexpr int SCAN_ITEMS = 8;
constexpr int SCAN_TILE = SCAN_THREADS * SCAN_ITEMS;

// exclusive scan of one tile per block, tile totals to sums[]
static __global__ void k_scan_tiles(const int *__restrict__ in, int *__restrict__ out, size_t n, int *__restrict__ sums) {
    __shared__ int wave_tot[SCAN_THREADS / WAVE];
    size_t base = (size_t)blockIdx.x * SCAN_TILE + (size_t)threadIdx.x * SCAN_ITEMS;
    int v[SCAN_ITEMS];
    int tsum = 0;
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; k++) {
        v[k] = (base + k < n) ? in[base + k] : 0;
        tsum += v[k];
    }
    // inclusive scan of tsum over the wave via DPP-free shuffles (ints, cheap), then across waves
    int lane = threadIdx.x & (WAVE - 1), wv = threadIdx.x / WAVE;
    int inc = tsum;
#pragma unroll
    for (int off = 1; off < WAVE; off <<= 1) {
        int t = __shfl_up(inc, off);
        if (lane >= off) inc += t;
    }
    if (lane == WAVE - 1) wave_tot[wv] = inc;
    __syncthreads();
    int wave_off = 0, total = 0;
#pragma unroll
    for (int w = 0; w < SCAN_THREADS / WAVE; w++) {
        if (w < wv) wave_off += wave_tot[w];
        total += wave_tot[w];
    }
    int run = wave_off + inc - tsum;
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; k++) {
        if (base + k < n) out[base + k] = run;
        run += v[k];
    }
    if (threadIdx.x == 0 && sums) sums[blockIdx.x] = total;
}

static __global__ void k_scan_add(int *__restrict__ out, size_t n, const int *__restrict__ offs) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] += offs[i / SCAN_TILE];
}

// One workgroup scans up to a few ten thousand values in place (4096 per trip: coalesced 16-byte loads, a block scan, a
// carry): the count arrays of the decomposition's partitions, for which three launches of the tiled scan were three launch
// latencies behind a blocking read-back.
constexpr int SCAN_BLOCK_MAX = 65536;
static __global__ __launch_bounds__(1024) void k_scan_block(int *__restrict__ data, int n) {
    __shared__ int wtot[1024 / WAVE];
    const int lane = threadIdx.x & (WAVE - 1), wv = threadIdx.x / WAVE;
    int carry = 0;
    for (int base = 0; base < n; base += 4096) {
        const int i0 = base + (int)threadIdx.x * 4;
        int v[4];
#pragma unroll
        for (int k = 0; k < 4; k++) v[k] = (i0 + k < n) ? data[i0 + k] : 0;
        const int s = v[0] + v[1] + v[2] + v[3];
        int inc = s;
#pragma unroll
        for (int off = 1; off < WAVE; off <<= 1) {
            const int t = __shfl_up(inc, off);
            if (lane >= off) inc += t;
        }
        if (lane == WAVE - 1) wtot[wv] = inc;
        __syncthreads();
        int woff = 0, tot = 0;
#pragma unroll
        for (int w = 0; w < 1024 / WAVE; w++) {
            if (w < wv) woff += wtot[w];
            tot += wtot[w];
        }
        int run = carry + woff + inc - s;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            if (i0 + k < n) data[i0 + k] = run;
            run += v[k];
        }
        carry += tot;
        __syncthreads();
    }
}

// Several small zero-fills in ONE launch (a hipMemsetAsync whose size is not a multiple of 16 bytes is two fill kernels,
// and every launch behind a blocking read-back is a launch latency on the critical path of a rebuild)
struct ZeroRanges {
    int *ptr[6];
    unsigned n[6];
};
static __global__ void k_zero_ranges(ZeroRanges z) {
    int *p = z.ptr[blockIdx.y];
    const unsigned n = z.n[blockIdx.y];
    for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) p[i] = 0;
}
struct Zeros {
    ZeroRanges z{};
    int count = 0;
    unsigned most = 0;
    Zeros &add(int *p, size_t n) {
        if (n == 0) return *this;
        z.ptr[count] = p; z.n[count] = (unsigned)n; count++;
        most = n > most ? (unsigned)n : most;
        return *this;
    }
    void run(hipStream_t s) {
        if (count == 0) return;
        const unsigned bx = most <= 256 ? 1u : (most + 1023u) / 1024u;
        hipLaunchKernelGGL(k_zero_ranges, dim3(bx > 512u ? 512u : bx, count), dim3(256), 0, s, z);
    }
};

// ------------------------------------------------------------------------------------ gathers
template <typename real>
__device__ __forceinline__ void store_rec(Rec<real> *rec, float *te, int p, real x, real y, real z, float hs, float tev);
template <>
__device__ __forceinline__ void store_rec<double>(Rec<double> *rec, float *, int p, double x, double y, double z,
                                                  float hs, float tev) {
    Rec<double> r;
    r.x = x; r.y = y; r.z = z; r.hs = hs; r.te = tev;
    rec[p] = r;
}
template <>
__device__ __forceinline__ void store_rec<float>(Rec<float> *rec, float *te, int p, float x, float y, float z, float hs,
                                                 float tev) {
    Rec<float> r;
    r.x = x; r.y = y; r.z = z; r.hs = hs;
    rec[p] = r;
    te[p] = tev;
}
__device__ __forceinline__ void rec_params(const Rec<double> *rec, const float *, int p, float &hs, float &tev) {
    hs = rec[p].hs; tev = rec[p].te;
}
__device__ __forceinline__ void rec_params(const Rec<float> *rec, const float *te, int p, float &hs, float &tev) {
    hs = rec[p].hs; tev = te[p];
}

// Cell-ordered records hold the image of each atom INSIDE the primary box along periodic
// dimensions (the LDS-tiled kernels add explicit +-L shifts and never apply a minimum image);
// the number of box lengths removed is kept per atom so caller-order copies return the
// caller's unwrapped coordinates.  Packed 3 x 10 bits, biased by 512.
constexpr int IMG_BIAS = 512;
__device__ __forceinline__ int img_pack(int kx, int ky, int kz) {
    return (kx + IMG_BIAS) | ((ky + IMG_BIAS) << 10) | ((kz + IMG_BIAS) << 20);
}
__device__ __forceinline__ void img_unpack(int v, int &kx, int &ky, int &kz) {
    kx = (v & 1023) - IMG_BIAS; ky = ((v >> 10) & 1023) - IMG_BIAS; kz = ((v >> 20) & 1023) - IMG_BIAS;
}
template <typename real>
__device__ __forceinline__ int wrap_into_box(real &p, real lo, real len, int periodic) {
    if (!periodic) return 0;
    const real k = floor((p - lo) / len);
    p -= k * len;
    return (int)k;
}

// cell-relative records: the periodic image of a coordinate (already relative to a cell origin) that lies next to that cell;
// returns the box lengths removed
__device__ __forceinline__ int rel_nearest_image(double &r, double len, int periodic) {
    if (!periodic) return 0;
    const double s = rint(r / len);
    r -= s * len;
    return (int)s;
}

// Build the cell-ordered state from caller-order arrays.  order[p] = caller id of slot p.
template <typename real>
__global__ void k_gather_user(int n, int n_owned, size_t pitch, GridP<real> g, const int *__restrict__ order,
                              const int *__restrict__ cell_of, const real *__restrict__ pos,
                              const emdee_lj_atom *__restrict__ atoms, const real *__restrict__ vel,
                              const real *__restrict__ inv_mass, Rec<real> *__restrict__ rec, float *__restrict__ te,
                              real *__restrict__ xb, real *__restrict__ v_out, real *__restrict__ im_out,
                              int *__restrict__ perm, int *__restrict__ inv_perm, int *__restrict__ cell_sorted,
                              int *__restrict__ img, int nt = 1, const long long *__restrict__ tag_in = nullptr,
                              long long *__restrict__ tag_out = nullptr, int ncell = 0, int *__restrict__ cstart = nullptr,
                              const int *__restrict__ tstart = nullptr, RelGrid rel_out = RelGrid{}) {
    int p = blockIdx.x * blockDim.x + threadIdx.x;
    // (typed boxes and x sub-bins: the first slot of every CELL from the per-(cell, digit) starts -- rides here instead of
    // being a launch of its own)
    if (cstart) for (int c = p; c <= ncell; c += gridDim.x * blockDim.x) cstart[c] = tstart[(size_t)c * nt];
    if (p >= n) return;
    int i = order[p];
    if (tag_out) tag_out[p] = tag_in[i];
    real x = pos[3 * (size_t)i], y = pos[3 * (size_t)i + 1], z = pos[3 * (size_t)i + 2];
    const int kx = wrap_into_box(x, g.lo[0], g.len[0], g.per[0]);
    const int ky = wrap_into_box(y, g.lo[1], g.len[1], g.per[1]);
    const int kz = wrap_into_box(z, g.lo[2], g.len[2], g.per[2]);
    img[p] = img_pack(kx, ky, kz);
    emdee_lj_atom a = atoms[i];
    if (sizeof(real) == 4 && rel_out.on) {                   // cell-relative records (RelGrid)
        double ox, oy, oz;
        rel_out.origin(cell_of[i] / nt, ox, oy, oz);
        double rx = (double)x - ox, ry = (double)y - oy, rz = (double)z - oz;
        int jx = kx + rel_nearest_image(rx, (double)g.len[0], g.per[0]), jy = ky + rel_nearest_image(ry, (double)g.len[1], g.per[1]),
            jz = kz + rel_nearest_image(rz, (double)g.len[2], g.per[2]);
        img[p] = img_pack(jx, jy, jz);
        x = (real)rx; y = (real)ry; z = (real)rz;
    }
    store_rec<real>(rec, te, p, x, y, z, a.half_sigma, a.twice_sqrt_eps);
    xb[p] = x; xb[pitch + p] = y; xb[2 * pitch + p] = z;
    bool owned = i < n_owned;
    if (v_out) {
        for (int d = 0; d < 3; d++) v_out[d * pitch + p] = (owned && vel) ? vel[3 * (size_t)i + d] : (real)0;
    }
    if (im_out) im_out[p] = (owned && inv_mass) ? inv_mass[i] : (real)1;
    perm[p] = i;
    inv_perm[i] = p;
    cell_sorted[p] = cell_of[i] / nt;
}

// Re-sort an already cell-ordered state (MD rebuild). order[p] = OLD slot of new slot p.
// EDIT (NbSystem::resort_edit, a decomposed domain's rebuild in its own order): the old slots are the ids from now on
// (perm[p] = o: atoms that stay keep no other name, arrivals and new ghosts sit behind the old state), the image counts
// start afresh (the positions were wrapped by the ownership pass; an atom that changes owner takes no count along), only
// the *n_dev items that were kept exist, and their number goes out with the words of the rebuild's one read-back.
template <typename real, bool EDIT = false>
__global__ void k_gather_sorted(int n, size_t pitch, GridP<real> g, const int *__restrict__ order,
                                const int *__restrict__ cell_of, const Rec<real> *__restrict__ rec_in,
                                const float *__restrict__ te_in, const real *__restrict__ v_in,
                                const real *__restrict__ im_in, const int *__restrict__ perm_in,
                                const int *__restrict__ img_in, Rec<real> *__restrict__ rec, float *__restrict__ te,
                                real *__restrict__ xb, real *__restrict__ v_out, real *__restrict__ im_out,
                                int *__restrict__ perm, int *__restrict__ inv_perm, int *__restrict__ cell_sorted,
                                int *__restrict__ img, int nt = 1, const long long *__restrict__ tag_in = nullptr,
                                long long *__restrict__ tag_out = nullptr, const int *__restrict__ n_dev = nullptr,
                                int *__restrict__ n_out = nullptr, int ncell = 0, int *__restrict__ cstart = nullptr,
                                const int *__restrict__ tstart = nullptr, RelGrid rel_in = RelGrid{}, RelGrid rel_out = RelGrid{}) {
    int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (cstart) for (int c = p; c <= ncell; c += gridDim.x * blockDim.x) cstart[c] = tstart[(size_t)c * nt];
    if (EDIT) {
        n = min(n, *n_dev);
        if (p == 0 && n_out) *n_out = *n_dev;
    }
    if (p >= n) return;
    int o = order[p];
    Rec<real> r = rec_in[o];
    float hs, tev;
    rec_params(rec_in, te_in, o, hs, tev);
    int kx = 0, ky = 0, kz = 0;
    if (!EDIT) img_unpack(img_in[o], kx, ky, kz);
    if (sizeof(real) == 4 && (rel_in.on || rel_out.on)) {
        // cell-relative records: the absolute position in double, wrapped, then relative to the NEW cell (the cell the
        // binning pass found from the same position rounded to fp32: should the rounding have crossed a cell boundary, the
        // relative coordinate is a hair outside [0, width) -- as after any step)
        double ax = (double)r.x, ay = (double)r.y, az = (double)r.z, ox, oy, oz;
        if (rel_in.on) { rel_in.origin(rel_in.cell[o], ox, oy, oz); ax += ox; ay += oy; az += oz; }
        kx += wrap_into_box(ax, (double)g.lo[0], (double)g.len[0], g.per[0]);
        ky += wrap_into_box(ay, (double)g.lo[1], (double)g.len[1], g.per[1]);
        kz += wrap_into_box(az, (double)g.lo[2], (double)g.len[2], g.per[2]);
        if (rel_out.on) {
            rel_out.origin(cell_of[o] / nt, ox, oy, oz); ax -= ox; ay -= oy; az -= oz;
            // At a periodic face the binning (position rounded to fp32) and the wrap above (double) may pick different images of
            // the same atom -- fp32(L - 1e-6) is L, i.e. cell 0, while the double stays below L: take the image next to the cell
            // the atom was binned into (found on the GPU: an atom at z = 0 came out a whole box away from its cell, with an empty row)
            kx += rel_nearest_image(ax, (double)g.len[0], g.per[0]);
            ky += rel_nearest_image(ay, (double)g.len[1], g.per[1]);
            kz += rel_nearest_image(az, (double)g.len[2], g.per[2]);
        }
        r.x = (real)ax; r.y = (real)ay; r.z = (real)az;
    } else {
        kx += wrap_into_box(r.x, g.lo[0], g.len[0], g.per[0]);
        ky += wrap_into_box(r.y, g.lo[1], g.len[1], g.per[1]);
        kz += wrap_into_box(r.z, g.lo[2], g.len[2], g.per[2]);
    }
    img[p] = img_pack(kx, ky, kz);
    store_rec<real>(rec, te, p, r.x, r.y, r.z, hs, tev);
    xb[p] = r.x; xb[pitch + p] = r.y; xb[2 * pitch + p] = r.z;
    if (v_out) {
        for (int d = 0; d < 3; d++) v_out[d * pitch + p] = v_in[d * pitch + o];
    }
    if (im_out) im_out[p] = im_in[o];
    if (tag_out) tag_out[p] = tag_in[o];
    int i = EDIT ? o : perm_in[o];
    perm[p] = i;
    inv_perm[i] = p;
    cell_sorted[p] = cell_of[o] / nt;
}

// Operator path: same list, new caller positions -> refresh the records in place, each atom in the
// periodic image nearest to where it was at build time (the caller may have wrapped or shifted it).
template <typename real>
__global__ void k_refresh_positions(int n, size_t pitch, GridP<real> g, const int *__restrict__ perm,
                                    const real *__restrict__ pos, const emdee_lj_atom *__restrict__ atoms,
                                    const real *__restrict__ xb, Rec<real> *__restrict__ rec, float *__restrict__ te) {
    int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    int i = perm[p];
    emdee_lj_atom a = atoms[i];
    const real bx = xb[p], by = xb[pitch + p], bz = xb[2 * pitch + p];
    const real x = bx + min_image(pos[3 * (size_t)i] - bx, g.plen[0], g.pinv[0]);
    const real y = by + min_image(pos[3 * (size_t)i + 1] - by, g.plen[1], g.pinv[1]);
    const real z = bz + min_image(pos[3 * (size_t)i + 2] - bz, g.plen[2], g.pinv[2]);
    store_rec<real>(rec, te, p, x, y, z, a.half_sigma, a.twice_sqrt_eps);
}

// Operator path, one pass per call: refresh the records (as k_refresh_positions), raise flags[1] if an atom has moved
// more than sqrt(thr2) since the build (the caller then rebuilds and this refresh is discarded) and flags[5] if the
// LJAtom array is not one value repeated (which selects between the single-species and the general kernels).
template <typename real>
__global__ void k_refresh_check(int n, size_t pitch, GridP<real> g, const int *__restrict__ perm, const real *__restrict__ pos,
                                const emdee_lj_atom *__restrict__ atoms, const real *__restrict__ xb, Rec<real> *__restrict__ rec,
                                float *__restrict__ te, real thr2, int *__restrict__ flags, int typed = 0) {
    int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    int i = perm[p];
    const emdee_lj_atom a = atoms[i], first = atoms[0];
    if (typed) {   // a typed box is SORTED by species: an atom whose LJAtom was edited invalidates the order and the list
        float hs0, te0;
        rec_params(rec, te, p, hs0, te0);
        if (__float_as_int(a.half_sigma) != __float_as_int(hs0) || __float_as_int(a.twice_sqrt_eps) != __float_as_int(te0)) flags[1] = 1;
    }
    const real bx = xb[p], by = xb[pitch + p], bz = xb[2 * pitch + p];
    const real dx = min_image(pos[3 * (size_t)i] - bx, g.plen[0], g.pinv[0]);
    const real dy = min_image(pos[3 * (size_t)i + 1] - by, g.plen[1], g.pinv[1]);
    const real dz = min_image(pos[3 * (size_t)i + 2] - bz, g.plen[2], g.pinv[2]);
    store_rec<real>(rec, te, p, bx + dx, by + dy, bz + dz, a.half_sigma, a.twice_sqrt_eps);
    if (dx * dx + dy * dy + dz * dz > thr2) flags[1] = 1;
    if (__float_as_int(a.half_sigma) != __float_as_int(first.half_sigma) ||
        __float_as_int(a.twice_sqrt_eps) != __float_as_int(first.twice_sqrt_eps))
        flags[5] = 1;
    // the first LJAtom itself rides along (flags[14], [15]): the host needs it for the single-species constants, and one
    // posted read-back of the words is cheaper than two copies and a synchronisation
    if (p == 0) { flags[14] = __float_as_int(first.half_sigma); flags[15] = __float_as_int(first.twice_sqrt_eps); }
}

// Operator path: has any atom moved more than sqrt(thr2) (minimum image) since the build?
template <typename real>
__global__ void k_check_displacement(int n, const int *__restrict__ inv_perm, const real *__restrict__ pos,
                                     const real *__restrict__ xb, size_t pitch, GridP<real> g, real thr2,
                                     int *__restrict__ flag) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int p = inv_perm[i];
    real dx = min_image(pos[3 * (size_t)i] - xb[p], g.plen[0], g.pinv[0]);
    real dy = min_image(pos[3 * (size_t)i + 1] - xb[pitch + p], g.plen[1], g.pinv[1]);
    real dz = min_image(pos[3 * (size_t)i + 2] - xb[2 * pitch + p], g.plen[2], g.pinv[2]);
    if (dx * dx + dy * dy + dz * dz > thr2) *flag = 1;
}

// ------------------------------------------------------------------------------------ neighbour build
// One wavefront per atom (cell order).  The stencil is walked as rows of cells along x: in cell
// order a row segment is ONE contiguous slot range, so the 64 lanes read 64 consecutive records
// (coalesced), test the minimum-image distance, and compact the survivors with ballot + mbcnt.
constexpr int NBR_BLOCK = 256;

template <typename real>
__global__ __launch_bounds__(NBR_BLOCK) void k_nbr_build(int n, int n_owned, AtomView<real> atoms,
                                                          const int *__restrict__ perm,
                                                          const int *__restrict__ cell_sorted,
                                                          const int *__restrict__ start, GridP<real> g, real rlist2,
                                                          int *__restrict__ nbr, int stride, int *__restrict__ cnt,
                                                          int *__restrict__ overflow) {
    const int lane = threadIdx.x & (WAVE - 1);
    const int p = __builtin_amdgcn_readfirstlane(blockIdx.x * (NBR_BLOCK / WAVE) + threadIdx.x / WAVE);
    if (p >= n) return;
    if (perm[p] >= n_owned) {   // ghosts act on owned atoms but own no row
        if (lane == 0) cnt[p] = 0;
        return;
    }
    real xi, yi, zi, hs_unused, te_unused;
    load_atom(atoms, p, xi, yi, zi, hs_unused, te_unused);
    const int c = cell_sorted[p];
    const int Mx = g.M[0], My = g.M[1], Mz = g.M[2], nd = g.nd;
    const int cx = c % Mx, cy = (c / Mx) % My, cz = c / (Mx * My);
    int *row = nbr + (size_t)p * stride;
    int count = 0;

    const int span = 2 * nd + 1;
    const int nz = g.per[2] ? min(span, Mz) : span;
    const int ny = g.per[1] ? min(span, My) : span;
    for (int kz = 0; kz < nz; kz++) {
        int zz = cz - nd + kz;
        if (g.per[2]) zz = ((zz % Mz) + Mz) % Mz;
        else if (zz < 0 || zz >= Mz) continue;
        for (int ky = 0; ky < ny; ky++) {
            int yy = cy - nd + ky;
            if (g.per[1]) yy = ((yy % My) + My) % My;
            else if (yy < 0 || yy >= My) continue;
            const int base = (zz * My + yy) * Mx;
            // x range of this row as one or two contiguous cell segments
            int xa0, xb0, xa1 = 0, xb1 = -1;
            int xa = cx - nd, xb = cx + nd;
            if (g.per[0]) {
                if (Mx <= span) { xa0 = 0; xb0 = Mx - 1; }
                else if (xa < 0) { xa0 = xa + Mx; xb0 = Mx - 1; xa1 = 0; xb1 = xb; }
                else if (xb >= Mx) { xa0 = xa; xb0 = Mx - 1; xa1 = 0; xb1 = xb - Mx; }
                else { xa0 = xa; xb0 = xb; }
            } else {
                xa0 = max(xa, 0); xb0 = min(xb, Mx - 1);
            }
            for (int seg = 0; seg < 2; seg++) {
                const int a = seg ? xa1 : xa0, b = seg ? xb1 : xb0;
                if (b < a) continue;
                const int s = start[base + a], e = start[base + b + 1];
                for (int q0 = s; q0 < e; q0 += WAVE) {
                    const int q = q0 + lane;
                    bool pass = false;
                    if (q < e && q != p) {
                        real xj, yj, zj, hs_j, te_j;
                        load_atom(atoms, q, xj, yj, zj, hs_j, te_j);
                        real dx = min_image(xi - xj, g.plen[0], g.pinv[0]);
                        real dy = min_image(yi - yj, g.plen[1], g.pinv[1]);
                        real dz = min_image(zi - zj, g.plen[2], g.pinv[2]);
                        pass = dx * dx + dy * dy + dz * dz < rlist2;
                    }
                    const unsigned long long mask = __ballot(pass);
                    if (pass) {
                        const int k = count + prefix_popc(mask);
                        if (k < stride) row[k] = q;
                    }
                    count += __popcll(mask);
                }
            }
        }
    }
    if (lane == 0) {
        cnt[p] = min(count, stride);
        if (count > stride) atomicMax(overflow, count);   // rare: host grows the stride and rebuilds
    }
}

// ------------------------------------------------------------------------------------ force kernel
// lj_force_nbr: one wavefront per atom.  Each block owns FORCE_ATOMS consecutive atoms (cell
// order), each of its 4 waves walks FORCE_ATOMS/4 of them: the 64 lanes read 64 neighbour
// indices per 256-B segment, gather one record each, evaluate the pair function, and the
// per-lane partial sums are combined with DPP row shifts/broadcasts.  Results are staged in
// LDS and stored once per block as unit-stride segments of the SoA force planes: no atomics,
// no pre-zeroing (the reference zero-fills and atomically accumulates, src/nonbonded.jl:88-104,
// 112-114).  Block ids are remapped so that each XCD (private L2) owns one contiguous span of
// atoms: neighbouring atoms gather the same records.
constexpr int FORCE_BLOCK = 256;
constexpr int FORCE_ATOMS = 64;
constexpr int NXCD = 8;

template <typename real, int BITMASK>
__global__ __launch_bounds__(FORCE_BLOCK) void k_lj_force_nbr(int n, int n_owned, int nblocks_per_xcd,
                                                              AtomView<real> atoms, const int *__restrict__ perm,
                                                              const int *__restrict__ nbr, int stride,
                                                              const int *__restrict__ cnt, GridP<real> g,
                                                              LJModel<real> model, size_t pitch,
                                                              real *__restrict__ frc, real *__restrict__ en,
                                                              real *__restrict__ vir, const int *__restrict__ guard = nullptr) {
    __shared__ real s_out[5][FORCE_ATOMS];
    if (guard != nullptr && *guard != 0) return;   // a step queued behind a rebuild request (emdee_dd_step): leave no trace
    const int lane = threadIdx.x & (WAVE - 1);
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x / WAVE);
    // XCD-aware remap: hardware deals consecutive block ids round-robin over the 8 XCDs
    const int lb = (blockIdx.x % NXCD) * nblocks_per_xcd + blockIdx.x / NXCD;
    const int first = lb * FORCE_ATOMS;
    if (first >= n) return;
    constexpr int PER_WAVE = FORCE_ATOMS / (FORCE_BLOCK / WAVE);

    for (int a = 0; a < PER_WAVE; a++) {
        const int slot = wv * PER_WAVE + a;
        const int p = first + slot;
        if (p >= n) break;
        real fx = 0, fy = 0, fz = 0, e = 0, w = 0;
        if (perm[p] < n_owned) {
            real xi, yi, zi, hs_i, te_i;
            load_atom(atoms, p, xi, yi, zi, hs_i, te_i);
            const int m = cnt[p];
            const int *row = nbr + (size_t)p * stride;
            for (int k = lane; k < m; k += WAVE) {
                const int j = row[k];
                real xj, yj, zj, hs_j, te_j;
                load_atom(atoms, j, xj, yj, zj, hs_j, te_j);
                const real dx = min_image(xi - xj, g.plen[0], g.pinv[0]);
                const real dy = min_image(yi - yj, g.plen[1], g.pinv[1]);
                const real dz = min_image(zi - zj, g.plen[2], g.pinv[2]);
                const real r2 = dx * dx + dy * dy + dz * dz;
                if (r2 < model.rc2) {   // strict test (Q2): listed-but-outside pairs cost nothing more
                    const real inv_r2 = fast_rcp(r2);
                    real E, W;
                    lj_interaction(r2, inv_r2, model, hs_i, te_i, hs_j, te_j, E, W);
                    if (BITMASK & EMDEE_FORCES) {
                        const real wr2 = W * inv_r2;   // src/nonbonded.jl:139
                        fx += wr2 * dx; fy += wr2 * dy; fz += wr2 * dz;
                    }
                    if (BITMASK & EMDEE_ENERGIES) e += E;
                    if (BITMASK & EMDEE_VIRIALS) w += W;
                }
            }
            if (BITMASK & EMDEE_FORCES) {
                fx = wave_sum_to_lane63(fx); fy = wave_sum_to_lane63(fy); fz = wave_sum_to_lane63(fz);
            }
            if (BITMASK & EMDEE_ENERGIES) e = wave_sum_to_lane63(e);
            if (BITMASK & EMDEE_VIRIALS) w = wave_sum_to_lane63(w);
        }
        if (lane == WAVE - 1) {
            s_out[0][slot] = fx; s_out[1][slot] = fy; s_out[2][slot] = fz;
            s_out[3][slot] = (real)0.5 * e;   // half of each pair term per atom, src/nonbonded.jl:142-145
            s_out[4][slot] = (real)0.5 * w;
        }
    }
    __syncthreads();
    const int t = threadIdx.x, col = t & (FORCE_ATOMS - 1), plane = t / FORCE_ATOMS;   // 4 planes of 64 threads
    if (first + col < n) {
        if ((BITMASK & EMDEE_FORCES) && plane < 3) frc[plane * pitch + first + col] = s_out[plane][col];
        if (plane == 3) {
            if (BITMASK & EMDEE_ENERGIES) en[first + col] = s_out[3][col];
            if (BITMASK & EMDEE_VIRIALS) vir[first + col] = s_out[4][col];
        }
    }
}

// ------------------------------------------------------------------------------------ integrator
// verlet_kick_drift: v += c f / m ; x += dt v, one pass (c = dt/2 for a lone half kick, dt when
// the closing half kick of the previous step is fused in), plus the rebuild trigger:
// |x - x_build|^2 > (skin/2)^2 raises *flag.
// noise != NULL: Langevin O step between the kick and the drift, v = c1 v + noise (k_langevin_noise).
template <typename real>
__global__ void k_kick_drift(int n, int n_owned, size_t pitch, const int *__restrict__ perm, Rec<real> *__restrict__ rec,
                             real *__restrict__ vel, const real *__restrict__ frc, const real *__restrict__ inv_mass,
                             real c, real dt, const real *__restrict__ xb, real thr2, int *__restrict__ flag,
                             const real *__restrict__ noise, real c1, const int *__restrict__ guard = nullptr,
                             int *__restrict__ far_word = nullptr, real thr2_near = (real)0) {
    int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (guard != nullptr && *guard != 0) {   // queued behind a rebuild request: do nothing and pass the request on
        if (p == 0) *flag = 1;
        return;
    }
    if (p >= n) return;
    if (perm[p] >= n_owned) return;
    Rec<real> r = rec[p];
    const real cm = inv_mass ? c * inv_mass[p] : c;
    real vx = vel[p] + cm * frc[p];
    real vy = vel[pitch + p] + cm * frc[pitch + p];
    real vz = vel[2 * pitch + p] + cm * frc[2 * pitch + p];
    if (noise) {
        vx = c1 * vx + noise[p]; vy = c1 * vy + noise[pitch + p]; vz = c1 * vz + noise[2 * pitch + p];
    }
    vel[p] = vx; vel[pitch + p] = vy; vel[2 * pitch + p] = vz;
    r.x += dt * vx; r.y += dt * vy; r.z += dt * vz;
    rec[p] = r;
    const real dx = r.x - xb[p], dy = r.y - xb[pitch + p], dz = r.z - xb[2 * pitch + p];
    if (dx * dx + dy * dy + dz * dz > thr2) *flag = 1;
    if (far_word && dx * dx + dy * dy + dz * dz > thr2_near) *far_word = 1;   // (brick.hpp BrickArgs::far_skip)
}

// ---- Langevin thermostat (build-defined; the reference has neither integrator nor thermostat) ----------
// Three N(0,1) numbers per (seed, step, atom id): splitmix64-finalised counters + Box-Muller, so the noise of
// an atom does not depend on where the sort put it or on which rank owns it.  Same integer arithmetic as
// oracle/emdee_oracle.c orc_langevin_normals.
__device__ __forceinline__ unsigned long long lgv_mix(unsigned long long z) {
    z ^= z >> 30; z *= 0xBF58476D1CE4E5B9ull;
    z ^= z >> 27; z *= 0x94D049BB133111EBull;
    z ^= z >> 31;
    return z;
}
__device__ __forceinline__ void lgv_normals(unsigned long long seed, unsigned long long step, unsigned long long id,
                                            double out[3]) {
    const unsigned long long base = lgv_mix(seed + 0x9E3779B97F4A7C15ull * id);
    const unsigned long long s2 = lgv_mix(base ^ (0xD1B54A32D192ED03ull * (step + 1)));
    unsigned long long r[4];
#pragma unroll
    for (int j = 0; j < 4; j++) r[j] = lgv_mix(s2 + 0x9E3779B97F4A7C15ull * (unsigned long long)(j + 1));
    const double two53 = 1.0 / 9007199254740992.0, twopi = 6.283185307179586476925286766559;
    const double u1 = (double)((r[0] >> 11) + 1) * two53, v2 = (double)(r[1] >> 11) * two53;
    const double u3 = (double)((r[2] >> 11) + 1) * two53, v4 = (double)(r[3] >> 11) * two53;
    const double a = sqrt(-2.0 * log(u1)), b = sqrt(-2.0 * log(u3));
    double sn, cs;
    sincos(twopi * v2, &sn, &cs);
    out[0] = a * cs;
    out[1] = a * sn;
    out[2] = b * cos(twopi * v4);
}
// noise[p] = c2 sqrt(T / m_p) xi(seed, step, id_p) in cell order, one thread per atom (the fused step kernel's
// owner lanes only pick the three numbers up: the transcendental work stays out of the pair kernel)
template <typename real>
__global__ void k_langevin_noise(int n, int n_owned, size_t pitch, const int *__restrict__ perm,
                                 const long long *__restrict__ ids, const real *__restrict__ inv_mass,
                                 unsigned long long seed, unsigned long long step, double c2, double temperature,
                                 real *__restrict__ noise, const long long *__restrict__ tag = nullptr) {
    int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    const int i = perm[p];
    if (i >= n_owned) return;
    double xi[3];
    // (tag: the ids that travel with the atoms in cell order -- decomposed domains; ids: a caller-order array)
    lgv_normals(seed, step, (unsigned long long)(tag ? tag[p] : (ids ? ids[i] : (long long)i)), xi);
    const double amp = c2 * sqrt(temperature * (inv_mass ? (double)inv_mass[p] : 1.0));
    noise[p] = (real)(amp * xi[0]); noise[pitch + p] = (real)(amp * xi[1]); noise[2 * pitch + p] = (real)(amp * xi[2]);
}
template <typename real>
__global__ void k_langevin_normals_test(int n, unsigned long long seed, unsigned long long step, const long long *ids,
                                        double *out) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double xi[3];
    lgv_normals(seed, step, (unsigned long long)ids[i], xi);
    out[3 * i] = xi[0]; out[3 * i + 1] = xi[1]; out[3 * i + 2] = xi[2];
}

// verlet_kick: v += c f / m
template <typename real>
__global__ void k_kick(int n, int n_owned, size_t pitch, const int *__restrict__ perm, real *__restrict__ vel,
                       const real *__restrict__ frc, const real *__restrict__ inv_mass, real c) {
    int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    if (perm[p] >= n_owned) return;
    const real cm = inv_mass ? c * inv_mass[p] : c;
    vel[p] += cm * frc[p];
    vel[pitch + p] += cm * frc[pitch + p];
    vel[2 * pitch + p] += cm * frc[2 * pitch + p];
}

// ------------------------------------------------------------------------------------ caller-order copies
// One thread per cell-order slot p, written to the atom's id i = perm[p] -- or, when the ids have gaps (a decomposed domain
// that re-sorted its own records: NbSystem::resort_edit), to the rank of that id among the ids in use, cmap[i]: owned atoms
// first, then the ghosts, as every caller of these copies expects.  raw: the record's own coordinates, without the box
// lengths the sorts removed (the ownership pass of a decomposition wraps them anyway).
template <typename real>
__global__ void k_unsort(int n_owned, int n_total, size_t pitch, GridP<real> g, const int *__restrict__ img,
                         const int *__restrict__ perm, const int *__restrict__ cmap,
                         const Rec<real> *__restrict__ rec, const float *__restrict__ te, const real *__restrict__ vel,
                         const real *__restrict__ frc,
                         const real *__restrict__ en, const real *__restrict__ vir, const long long *__restrict__ tag,
                         real *__restrict__ pos_out,
                         real *__restrict__ vel_out, real *__restrict__ frc_out, real *__restrict__ en_out,
                         real *__restrict__ vir_out, emdee_lj_atom *__restrict__ atoms_out, long long *__restrict__ tag_out,
                         int raw, RelGrid rel = RelGrid{}) {
    int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n_total) return;
    const int id = perm[p];
    const int i = cmap ? cmap[id] : id;
    if (pos_out) {
        Rec<real> r = rec[p];
        int kx = 0, ky = 0, kz = 0;
        if (!raw) img_unpack(img[p], kx, ky, kz);   // back to the caller's (unwrapped) image
        if (sizeof(real) == 4 && rel.on) {          // cell-relative records: origin + record + images, in double, rounded once
            double ox, oy, oz;
            rel.origin(rel.cell[p], ox, oy, oz);
            pos_out[3 * (size_t)i] = (real)((double)r.x + ox + (double)kx * (double)g.len[0]);
            pos_out[3 * (size_t)i + 1] = (real)((double)r.y + oy + (double)ky * (double)g.len[1]);
            pos_out[3 * (size_t)i + 2] = (real)((double)r.z + oz + (double)kz * (double)g.len[2]);
        } else {
            pos_out[3 * (size_t)i] = r.x + (real)kx * g.len[0];
            pos_out[3 * (size_t)i + 1] = r.y + (real)ky * g.len[1];
            pos_out[3 * (size_t)i + 2] = r.z + (real)kz * g.len[2];
        }
    }
    if (atoms_out) {
        emdee_lj_atom a;
        rec_params(rec, te, p, a.half_sigma, a.twice_sqrt_eps);
        atoms_out[i] = a;
    }
    if (tag_out) tag_out[i] = tag[p];
    if (id >= n_owned) return;
    if (vel_out) for (int d = 0; d < 3; d++) vel_out[3 * (size_t)i + d] = vel[d * pitch + p];
    if (frc_out) for (int d = 0; d < 3; d++) frc_out[3 * (size_t)i + d] = frc[d * pitch + p];
    if (en_out) en_out[i] = en[p];
    if (vir_out) vir_out[i] = vir[p];
}

// ids in use -> their ranks (cmap): live[id] = 1 for every id some slot carries, then an exclusive scan of live[]
static __global__ void k_mark_live(int n, const int *__restrict__ perm, int *__restrict__ live) {
    int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p < n) live[perm[p]] = 1;
}

// sets *flag if any LJAtom differs (bitwise) from the first one
static __global__ void k_atoms_differ(int n, const emdee_lj_atom *__restrict__ atoms, int *__restrict__ flag) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const emdee_lj_atom a = atoms[i], b = atoms[0];
    if (__float_as_int(a.half_sigma) != __float_as_int(b.half_sigma) ||
        __float_as_int(a.twice_sqrt_eps) != __float_as_int(b.twice_sqrt_eps))
        *flag = 1;
}

// fused decomposed step: ghost records (not owned, never integrated here) follow the buffer swap
template <typename real>
__global__ void k_copy_ghost_records(int n, int n_owned, const int *__restrict__ perm, const Rec<real> *__restrict__ src,
                                     Rec<real> *__restrict__ dst) {
    int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p < n && perm[p] >= n_owned) dst[p] = src[p];
}

// halo exchange helpers: gather positions (+ periodic shift) of listed caller ids;
// scatter received positions into ghost records
template <typename real>
struct ShiftTable {
    real s[27][3];
};

template <typename real>
__global__ void k_pack_positions(int n, const int *__restrict__ ids, const int *__restrict__ codes, int n_shifts,
                                 const int *__restrict__ inv_perm, const Rec<real> *__restrict__ rec,
                                 ShiftTable<real> tab, real *__restrict__ buf) {
    int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    Rec<real> r = rec[inv_perm[ids[k]]];
    int c = codes ? codes[k] : 0;
    c = min(max(c, 0), n_shifts - 1);
    buf[3 * (size_t)k] = r.x + tab.s[c][0]; buf[3 * (size_t)k + 1] = r.y + tab.s[c][1]; buf[3 * (size_t)k + 2] = r.z + tab.s[c][2];
}

template <typename real>
__global__ void k_unpack_ghosts(int n, int first_id, const int *__restrict__ inv_perm, const real *__restrict__ buf,
                                Rec<real> *__restrict__ rec) {
    int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    int p = inv_perm[first_id + k];
    Rec<real> r = rec[p];
    r.x = buf[3 * (size_t)k]; r.y = buf[3 * (size_t)k + 1]; r.z = buf[3 * (size_t)k + 2];
    rec[p] = r;
}

// neighbour rows of the direct (int32, cell-order slot) list as caller ids
static __global__ void k_export_rows(int n, int n_owned, const int *__restrict__ perm, const int *__restrict__ nbr, int stride,
                                     const int *__restrict__ cnt, int *__restrict__ counts, int *__restrict__ out, int capacity,
                                     const int *__restrict__ cmap = nullptr) {
    int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    if (perm[p] >= n_owned) return;
    const int i = cmap ? cmap[perm[p]] : perm[p];
    const int m = cnt[p];
    counts[i] = m;
    for (int e = 0; e < m && e < capacity; e++) {
        const int j = perm[nbr[(size_t)p * stride + e]];
        out[(size_t)i * capacity + e] = cmap ? cmap[j] : j;
    }
}

// ------------------------------------------------------------------------------------ exclusions and 1-4 pairs
// (the hooks of src/modelling.jl:197-200.  The reference parses lj14scale and nothing consumes it --
// its hot path sums every pair, src/nonbonded.jl:129-150 -- so these are build-defined: pairs named by the caller are struck
// from the neighbour rows right after every build (the pair loop gets no mask), and the 1-4 pairs among them are
// evaluated on their own, scaled.)  Both tables are symmetric CSR lists over caller ids, partners ascending.
__device__ __forceinline__ bool csr_holds(const int *__restrict__ idx, int lo, int hi, int j) {
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        const int v = idx[mid];
        if (v == j) return true;
        if (v < j) lo = mid + 1; else hi = mid;
    }
    return false;
}

// rows of the direct (int32, cell-order slot) list without their excluded entries
static __global__ void k_filter_rows(int n, int n_owned, const int *__restrict__ perm, int *__restrict__ nbr, int stride,
                                     int *__restrict__ cnt, const int *__restrict__ ex_start, const int *__restrict__ ex_idx) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    const int i = perm[p];
    if (i >= n_owned) return;
    const int lo = ex_start[i], hi = ex_start[i + 1];
    if (lo == hi) return;
    int *row = nbr + (size_t)p * stride;
    const int m = min(cnt[p], stride);
    int w = 0;
    for (int e = 0; e < m; e++) {
        const int q = row[e];
        if (!csr_holds(ex_idx, lo, hi, perm[q])) row[w++] = q;
    }
    cnt[p] = w;
}

// 1-4 pairs: owner-computes over the symmetric table (no atomics, a fixed order of summation); the scaled pair terms are
// ADDED to what the list kernels have left -- in the cell-ordered arrays, or in the caller's arrays when the operator
// path had its results written there (user_*: caller order).
template <typename real>
__global__ void k_pairs14(int n, int n_owned, size_t pitch, AtomView<real> atoms, const int *__restrict__ perm,
                          const int *__restrict__ inv_perm, GridP<real> g, LJModel<real> model, const int *__restrict__ start14,
                          const int *__restrict__ idx14, real scale, int bitmask, real *__restrict__ frc, real *__restrict__ en,
                          real *__restrict__ vir, real *__restrict__ user_f, real *__restrict__ user_e, real *__restrict__ user_w) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    const int i = perm[p];
    if (i >= n_owned) return;
    const int lo = start14[i], hi = start14[i + 1];
    if (lo == hi) return;
    real xi, yi, zi, hs_i, te_i;
    load_atom(atoms, p, xi, yi, zi, hs_i, te_i);
    real fx = 0, fy = 0, fz = 0, e = 0, w = 0;
    for (int k = lo; k < hi; k++) {
        const int q = inv_perm[idx14[k]];
        real xj, yj, zj, hs_j, te_j;
        load_atom(atoms, q, xj, yj, zj, hs_j, te_j);
        const real dx = min_image(xi - xj, g.plen[0], g.pinv[0]);
        const real dy = min_image(yi - yj, g.plen[1], g.pinv[1]);
        const real dz = min_image(zi - zj, g.plen[2], g.pinv[2]);
        const real r2 = dx * dx + dy * dy + dz * dz;
        if (r2 < model.rc2) {                                  // CUTOFF semantics, as the list kernels
            const real inv_r2 = (real)1 / r2;
            real E, W;
            lj_interaction(r2, inv_r2, model, hs_i, te_i, hs_j, te_j, E, W);
            const real wr2 = W * inv_r2;                      // src/nonbonded.jl:139
            fx += wr2 * dx; fy += wr2 * dy; fz += wr2 * dz;
            e += E; w += W;
        }
    }
    fx *= scale; fy *= scale; fz *= scale;
    e *= (real)0.5 * scale; w *= (real)0.5 * scale;           // src/nonbonded.jl:142-145: half of a pair's E and W to either atom
    if (user_f != nullptr || user_e != nullptr || user_w != nullptr) {
        if ((bitmask & EMDEE_FORCES) && user_f) { user_f[3 * (size_t)i] += fx; user_f[3 * (size_t)i + 1] += fy; user_f[3 * (size_t)i + 2] += fz; }
        if ((bitmask & EMDEE_ENERGIES) && user_e) user_e[i] += e;
        if ((bitmask & EMDEE_VIRIALS) && user_w) user_w[i] += w;
    } else {
        if (bitmask & EMDEE_FORCES) { frc[p] += fx; frc[pitch + p] += fy; frc[2 * pitch + p] += fz; }
        if (bitmask & EMDEE_ENERGIES) en[p] += e;
        if (bitmask & EMDEE_VIRIALS) vir[p] += w;
    }
}

// ------------------------------------------------------------------------------------ reductions
// Deterministic two-stage sums in fp64 (mixed precision: fp32 per-atom values, fp64 totals).
constexpr int RED_BLOCK = 256;
constexpr int RED_MAX_BLOCKS = 1024;

__device__ __forceinline__ double block_sum(double v, double *sh) {
    v = wave_sum_to_lane63(v);
    const int lane = threadIdx.x & (WAVE - 1), wv = threadIdx.x / WAVE;
    if (lane == WAVE - 1) sh[wv] = v;
    __syncthreads();
    double t = 0.0;
    if (threadIdx.x == 0) for (int w = 0; w < RED_BLOCK / WAVE; w++) t += sh[w];
    __syncthreads();
    return t;   // valid in thread 0
}

// partial[b][0..2] = sum e, kinetic energy (with optional pending half kick c f/m), sum w
template <typename real>
__global__ __launch_bounds__(RED_BLOCK) void k_energy_partials(int n, int n_owned, size_t pitch,
                                                                const int *__restrict__ perm,
                                                                const real *__restrict__ en, const real *__restrict__ vir,
                                                                const real *__restrict__ vel, const real *__restrict__ frc,
                                                                const real *__restrict__ inv_mass, real c,
                                                                double *__restrict__ partial) {
    __shared__ double sh[RED_BLOCK / WAVE];
    double se = 0.0, sk = 0.0, sw = 0.0;
    for (int p = blockIdx.x * RED_BLOCK + threadIdx.x; p < n; p += gridDim.x * RED_BLOCK) {
        if (perm[p] >= n_owned) continue;
        if (en) se += (double)en[p];
        if (vir) sw += (double)vir[p];
        if (vel) {
            const real im = inv_mass ? inv_mass[p] : (real)1;
            double k2 = 0.0;
            for (int d = 0; d < 3; d++) {
                double v = (double)vel[d * pitch + p] + (double)(c * im) * (double)frc[d * pitch + p];
                k2 += v * v;
            }
            sk += 0.5 * k2 / (double)im;
        }
    }
    double t;
    t = block_sum(se, sh); if (threadIdx.x == 0) partial[3 * blockIdx.x] = t;
    t = block_sum(sk, sh); if (threadIdx.x == 0) partial[3 * blockIdx.x + 1] = t;
    t = block_sum(sw, sh); if (threadIdx.x == 0) partial[3 * blockIdx.x + 2] = t;
}

static __global__ __launch_bounds__(RED_BLOCK) void k_final_sum3(int nblocks, const double *__restrict__ partial,
                                                          double *__restrict__ out) {
    __shared__ double sh[RED_BLOCK / WAVE];
    for (int q = 0; q < 3; q++) {
        double s = 0.0;
        for (int b = threadIdx.x; b < nblocks; b += RED_BLOCK) s += partial[3 * b + q];
        double t = block_sum(s, sh);
        if (threadIdx.x == 0) out[q] = t;
    }
}

// list statistics: out[0] = entries, out[1] = max row, out[2] = entries with r2 < rc2
template <typename real>
__global__ __launch_bounds__(RED_BLOCK) void k_list_stats(int n, AtomView<real> atoms, const int *__restrict__ nbr,
                                                          int stride, const int *__restrict__ cnt, GridP<real> g,
                                                          real rc2, int count_pairs,
                                                          unsigned long long *__restrict__ out) {
    unsigned long long entries = 0, inside = 0;
    int mx = 0;
    for (int p = blockIdx.x * RED_BLOCK + threadIdx.x; p < n; p += gridDim.x * RED_BLOCK) {
        int m = cnt[p];
        entries += (unsigned long long)m;
        mx = max(mx, m);
        if (count_pairs) {
            real xi, yi, zi, h, t;
            load_atom(atoms, p, xi, yi, zi, h, t);
            for (int k = 0; k < m; k++) {
                real xj, yj, zj;
                load_atom(atoms, nbr[(size_t)p * stride + k], xj, yj, zj, h, t);
                real dx = min_image(xi - xj, g.plen[0], g.pinv[0]);
                real dy = min_image(yi - yj, g.plen[1], g.pinv[1]);
                real dz = min_image(zi - zj, g.plen[2], g.pinv[2]);
                inside += (dx * dx + dy * dy + dz * dz < rc2) ? 1ull : 0ull;
            }
        }
    }
    atomicAdd(&out[0], entries);   // one integer atomic per thread of a <=1024-block grid: exact, order-free
    atomicMax(&out[1], (unsigned long long)mx);
    atomicAdd(&out[2], inside);
}

}  // namespace emdee
