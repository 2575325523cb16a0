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
// Work split inside a workgroup: groups of G lanes share one atom (G = 16: one DPP row per
// atom, 4 atoms per wavefront).  Lane l of the group takes neighbours l, l+G, l+2G, ...; the
// row is stored LANE-MAJOR in blocks of 8 G entries so that those are 8 consecutive uint16 =
// ONE 16-byte load per lane (256 contiguous bytes per atom for G = 16), issued one atom ahead of
// the arithmetic.  Per-lane partial sums are combined with DPP row shifts (+ row broadcasts
// for G > 16).  Owner-computes, full list: no atomics, no pre-zeroing.
//
// Reference lines restated: pair function src/lennard_jones.jl:25-42 (lj_pair.hpp);
// f_ij = W/r2 * r_ij and the half split of E, W per atom src/nonbonded.jl:136-145; the cell
// grid convention src/cells.jl:82-85.  What replaces what: compute_tile! (src/nonbonded.jl:44-107)
// and find_action_partners1! (src/cells.jl:224-297).
#pragma once

#include "kernels.hpp"

namespace emdee {

enum BrickMode { BRICK_BUILD = 0, BRICK_FORCE = 1, BRICK_STATS = 2 };

constexpr int EPL = 8;   // neighbour entries per lane per 16-byte load

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
};

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
    int n, n_owned;
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
    LJModel<real> model;
    size_t pitch;
    real *frc, *en, *vir;
    unsigned long long *stats; // BRICK_STATS: [0] entries, [1] max row, [2] in-cutoff entries
};

// bytes of dynamic LDS k_brick needs
template <typename real, class Shape, int THREADS>
static inline size_t brick_lds_bytes(int tile_cap, int own_cap) {
    size_t tile_bytes = (size_t)tile_cap * sizeof(Rec<real>);
    size_t te_bytes = sizeof(real) == 4 ? (((size_t)tile_cap * 4 + 15) & ~(size_t)15) : 0;
    size_t ints = (Shape::NTC + 4) + Shape::NTC + Shape::NTC + (Shape::NOC + 4) + THREADS / WAVE + 2 + 2 * (size_t)own_cap;
    return tile_bytes + te_bytes + ((ints * 4 + 15) & ~(size_t)15);
}

// position of the e-th neighbour inside a row: blocks of 8 G entries, lane-major inside a block,
// so lane l of the group finds entries l, l+G, ..., l+7G of the block in 8 consecutive uint16
template <int G>
__host__ __device__ __forceinline__ int row_position(int e) {
    constexpr int BLK = EPL * G;
    const int blk = e / BLK, r = e % BLK;
    return blk * BLK + (r % G) * EPL + (r / G);
}

template <int G>
__device__ __forceinline__ unsigned long long group_bits(unsigned long long mask, int lane) {
    if (G == 64) return mask;
    const int base = lane & ~(G - 1);
    return (mask >> base) & ((1ull << G) - 1ull);
}

// max over the groups of a wavefront of a group-uniform value, as a scalar
template <int G>
__device__ __forceinline__ int wave_group_max(int v) {
#pragma unroll
    for (int off = G; off < WAVE; off <<= 1) v = max(v, __shfl_xor(v, off));
    return __builtin_amdgcn_readfirstlane(v);
}

__device__ __forceinline__ int pick16(const uint4 &q, int t) {   // t is a compile-time constant after unrolling
    const unsigned w = (t < 2) ? q.x : (t < 4) ? q.y : (t < 6) ? q.z : q.w;
    return (int)((t & 1) ? (w >> 16) : (w & 0xffffu));
}

template <typename real, class Shape, int THREADS, int G, int MODE, int BITMASK>
__global__ __launch_bounds__(THREADS) void k_brick(BrickArgs<real> a) {
    constexpr int BX = Shape::BX, BY = Shape::BY, TX = Shape::TX, TY = Shape::TY, NTC = Shape::NTC, NOC = Shape::NOC;
    constexpr int NWAVES = THREADS / WAVE;
    constexpr int GROUPS_PER_WAVE = WAVE / G;
    constexpr int NGROUPS = NWAVES * GROUPS_PER_WAVE;
    constexpr int BLK = EPL * G;

    // All LDS lives in the dynamic region with 16-byte carve offsets (a static __shared__ in front
    // would shift the base and put the ds_read_b128 gathers off their natural alignment).
    extern __shared__ __attribute__((aligned(16))) unsigned char s_dyn[];
    Rec<real> *tile = reinterpret_cast<Rec<real> *>(s_dyn);
    const size_t tile_bytes = (size_t)a.tile_cap * sizeof(Rec<real>);
    const size_t te_bytes = sizeof(real) == 4 ? (((size_t)a.tile_cap * 4 + 15) & ~(size_t)15) : 0;
    float *tile_te = reinterpret_cast<float *>(s_dyn + tile_bytes);   // fp32 only
    int *s_int = reinterpret_cast<int *>(s_dyn + tile_bytes + te_bytes);
    int *s_off = s_int;                    // [NTC+1] tile-local first slot of each tile cell
    int *s_gbeg = s_off + (NTC + 4);       // [NTC]   global (cell-order) first slot of each tile cell
    int *s_shift = s_gbeg + NTC;           // [NTC]   periodic image: 2 bits per dimension (0:-1, 1:0, 2:+1)
    int *s_own = s_shift + NTC;            // [NOC+1] prefix of own-cell populations
    int *s_wtot = s_own + (NOC + 4);       // [NWAVES]
    // [own_cap] per own atom {cell-order slot p, (row length or active flag) << 16 | tile slot}
    int2 *s_oinfo = reinterpret_cast<int2 *>(s_wtot + ((NWAVES + 1) & ~1));

    const int tid = threadIdx.x, lane = tid & (WAVE - 1), wv = tid / WAVE;
    const int lb = (blockIdx.x % NXCD) * a.bg.per_xcd + blockIdx.x / NXCD;   // XCD-contiguous brick order
    if (lb >= a.bg.nbricks) return;
    const int bxi = lb % a.bg.nb[0], byi = (lb / a.bg.nb[0]) % a.bg.nb[1], bzi = lb / (a.bg.nb[0] * a.bg.nb[1]);
    const int Mx = a.g.M[0], My = a.g.M[1], Mz = a.g.M[2];

    // ---- 1. tile cell table: population, global begin, periodic image --------------------------
    int my_cnt = 0;
    if (tid < NTC) {
        const int tx = tid % TX, ty = (tid / TX) % TY, tz = tid / (TX * TY);
        int gx = bxi * BX - 1 + tx, gy = byi * BY - 1 + ty, gz = bzi * Shape::BZ - 1 + tz;
        // own range may be clipped on the high side of a partial brick; halo = own range +- 1
        const int ox1 = min(bxi * BX + BX, Mx), oy1 = min(byi * BY + BY, My), oz1 = min(bzi * Shape::BZ + Shape::BZ, Mz);
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
        s_gbeg[tid] = gb;
        s_shift[tid] = sh;
    }
    {   // exclusive scan of my_cnt over the first NTC threads (NTC <= THREADS)
        int inc = my_cnt;
#pragma unroll
        for (int off = 1; off < WAVE; off <<= 1) {
            int t = __shfl_up(inc, off);
            if (lane >= off) inc += t;
        }
        if (lane == WAVE - 1) s_wtot[wv] = inc;
        __syncthreads();
        int woff = 0;
        for (int w = 0; w < wv; w++) woff += s_wtot[w];
        if (tid < NTC) s_off[tid] = woff + inc - my_cnt;
        if (tid == NTC - 1) s_off[NTC] = woff + inc;
    }
    __syncthreads();
    const int tile_n = s_off[NTC];
    if (tid == 0) {
        int acc = 0;
        for (int oc = 0; oc < NOC; oc++) {
            const int ox = oc % BX, oy = (oc / BX) % BY, oz = oc / (BX * BY);
            const int tc = (ox + 1) + TX * ((oy + 1) + TY * (oz + 1));
            s_own[oc] = acc;
            // cells past the box edge of a partial brick hold halo images, not atoms of this brick
            const bool mine = (bxi * BX + ox < Mx) && (byi * BY + oy < My) && (bzi * Shape::BZ + oz < Mz);
            acc += mine ? (s_off[tc + 1] - s_off[tc]) : 0;
        }
        s_own[NOC] = acc;
    }
    __syncthreads();
    const int n_own = s_own[NOC];
    if (tile_n > a.tile_cap || n_own > a.own_cap) {   // cannot happen: the host sized both from k_brick_tile_max
        if (tid == 0) atomicMax(&a.flags[2], max(tile_n, n_own));
        return;
    }

    // own atom o -> (tile slot, cell-order slot)
    auto locate = [&](int o, int &ti, int &p) {
        int oc = 0;
#pragma unroll
        for (int q = 1; q < NOC; q++) oc += (s_own[q] <= o) ? 1 : 0;
        const int ox = oc % BX, oy = (oc / BX) % BY, oz = oc / (BX * BY);
        const int tc = (ox + 1) + TX * ((oy + 1) + TY * (oz + 1));
        const int kk = o - s_own[oc];
        ti = s_off[tc] + kk;
        p = s_gbeg[tc] + kk;
        return oc;
    };

    // ---- 2. stage the tile: HBM -> LDS, unit stride inside each cell run, image shift applied ---
    for (int s = tid; s < tile_n; s += THREADS) {
        int lo = 0, hi = NTC;   // largest tc with s_off[tc] <= s
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (s_off[mid] <= s) lo = mid; else hi = mid;
        }
        const int gp = s_gbeg[lo] + (s - s_off[lo]);
        const int sh = s_shift[lo];
        Rec<real> r = a.rec[gp];
        r.x += (real)((sh & 3) - 1) * a.g.len[0];
        r.y += (real)(((sh >> 2) & 3) - 1) * a.g.len[1];
        r.z += (real)(((sh >> 4) & 3) - 1) * a.g.len[2];
        tile[s] = r;
        if (sizeof(real) == 4) tile_te[s] = a.te[gp];
    }
    // per own atom: where it lives and its row length (0 for ghosts: they own no row, get no force)
    for (int o = tid; o < n_own; o += THREADS) {
        int ti, p;
        locate(o, ti, p);
        const bool act = a.perm[p] < a.n_owned;
        const int m = (MODE == BRICK_BUILD) ? (act ? 1 : 0) : (act ? a.cnt[p] : 0);
        s_oinfo[o] = make_int2(p, (m << 16) | ti);
    }
    __syncthreads();

    // ---- 3. own atoms: one G-lane group per atom ------------------------------------------------
    const int gl = lane & (G - 1);                       // lane inside the group
    const int gid = wv * GROUPS_PER_WAVE + lane / G;     // group inside the block
    unsigned long long st_entries = 0, st_inside = 0;
    int st_max = 0;

    if (MODE == BRICK_BUILD) {
        for (int ob = 0; ob < n_own; ob += NGROUPS) {    // wave-uniform trip count
            const int o = ob + gid;
            const bool have = o < n_own;
            int ti = 0, p = 0, oc = 0;
            if (have) oc = locate(o, ti, p);
            const bool act = have && (s_oinfo[o].y >> 16) != 0;
            const int ox = oc % BX, oy = (oc / BX) % BY, oz = oc / (BX * BY);
            real xi, yi, zi, hs_i, te_i;
            tile_load<real>(tile, tile_te, ti, xi, yi, zi, hs_i, te_i);
            unsigned short *row = a.nbr + (size_t)p * a.stride;
            int count = 0;
#pragma unroll 1
            for (int dz = -1; dz <= 1; dz++) {
#pragma unroll 1
                for (int dy = -1; dy <= 1; dy++) {
                    const int tcr = ox + TX * ((oy + 1 + dy) + TY * (oz + 1 + dz));   // cell x-1 of that tile row
                    const int c0 = s_off[tcr];
                    const int span = act ? s_off[tcr + 3] - c0 : 0;                     // cells x-1, x, x+1: contiguous
                    const int wspan = wave_group_max<G>(span);
                    for (int cb = 0; cb < wspan; cb += G) {
                        const int c = c0 + cb + gl;
                        bool pass = false;
                        if (cb + gl < span && c != ti) {
                            real xj, yj, zj, hj, tj;
                            tile_load<real>(tile, tile_te, c, xj, yj, zj, hj, tj);
                            const real dx = xi - xj, dy2 = yi - yj, dz2 = zi - zj;
                            pass = dx * dx + dy2 * dy2 + dz2 * dz2 < a.rlist2;
                        }
                        const unsigned long long bits = group_bits<G>(__ballot(pass), lane);
                        if (pass) {
                            const int e = count + __popcll(bits & ((1ull << gl) - 1ull));
                            if (e < a.stride) row[row_position<G>(e)] = (unsigned short)c;
                        }
                        count += __popcll(bits);
                    }
                }
            }
            if (have && gl == 0) {
                a.cnt[p] = act ? min(count, a.stride) : 0;
                if (count > a.stride) atomicMax(&a.flags[0], count);
            }
        }
        return;
    }

    // FORCE / STATS: indices of the NEXT atom are fetched while the current one is being computed
    auto fetch = [&](int o) {
        uint4 q = make_uint4(0, 0, 0, 0);
        if (o < n_own) q = *reinterpret_cast<const uint4 *>(a.nbr + (size_t)s_oinfo[o].x * a.stride + gl * EPL);
        return q;
    };
    uint4 nxt = fetch(gid);
    for (int ob = 0; ob < n_own; ob += NGROUPS) {        // wave-uniform trip count
        const int o = ob + gid;
        const bool have = o < n_own;
        const int2 info = have ? s_oinfo[o] : make_int2(0, 0);
        const int p = info.x, ti = info.y & 0xffff, m = (int)((unsigned)info.y >> 16);
        uint4 cur = nxt;
        nxt = fetch(o + NGROUPS);
        const int wm = wave_group_max<G>(m);
        real xi, yi, zi, hs_i, te_i;
        tile_load<real>(tile, tile_te, ti, xi, yi, zi, hs_i, te_i);
        real fx = 0, fy = 0, fz = 0, e = 0, w = 0;
        for (int b0 = 0; b0 < wm; b0 += BLK) {
            if (b0 > 0) {   // rows longer than one block (rare): synchronous reload
                cur = make_uint4(0, 0, 0, 0);
                if (b0 < m) cur = *reinterpret_cast<const uint4 *>(a.nbr + (size_t)p * a.stride + b0 + gl * EPL);
            }
#pragma unroll
            for (int t = 0; t < EPL; t++) {
                if (b0 + t * G >= wm) break;             // wave-uniform
                if (b0 + t * G + gl < m) {
                    const int sj = pick16(cur, t);
                    real xj, yj, zj, hs_j, te_j;
                    tile_load<real>(tile, tile_te, sj, xj, yj, zj, hs_j, te_j);
                    const real dx = xi - xj, dy = yi - yj, dz = zi - zj;
                    const real r2 = dx * dx + dy * dy + dz * dz;
                    if (MODE == BRICK_STATS) {
                        st_inside += (r2 < a.model.rc2) ? 1ull : 0ull;
                    } else if (r2 < a.model.rc2) {       // strict test (Q2)
                        const real inv_r2 = fast_rcp(r2);
                        real E, W;
                        lj_interaction(r2, inv_r2, a.model, hs_i, te_i, hs_j, te_j, E, W);
                        if (BITMASK & EMDEE_FORCES) {
                            const real wr2 = W * inv_r2;   // src/nonbonded.jl:139
                            fx += wr2 * dx; fy += wr2 * dy; fz += wr2 * dz;
                        }
                        if (BITMASK & EMDEE_ENERGIES) e += E;
                        if (BITMASK & EMDEE_VIRIALS) w += W;
                    }
                }
            }
        }
        if (MODE == BRICK_STATS) {
            if (gl == 0) { st_entries += (unsigned long long)m; st_max = max(st_max, m); }
        } else {
            // all lanes are active here: DPP reductions see every lane of the group
            if (BITMASK & EMDEE_FORCES) {
                fx = group_sum_to_last<G>(fx); fy = group_sum_to_last<G>(fy); fz = group_sum_to_last<G>(fz);
            }
            if (BITMASK & EMDEE_ENERGIES) e = group_sum_to_last<G>(e);
            if (BITMASK & EMDEE_VIRIALS) w = group_sum_to_last<G>(w);
            if (have && gl == G - 1) {
                if (BITMASK & EMDEE_FORCES) {
                    a.frc[p] = fx; a.frc[a.pitch + p] = fy; a.frc[2 * a.pitch + p] = fz;
                }
                if (BITMASK & EMDEE_ENERGIES) a.en[p] = (real)0.5 * e;   // src/nonbonded.jl:142-145
                if (BITMASK & EMDEE_VIRIALS) a.vir[p] = (real)0.5 * w;
            }
        }
    }
    if (MODE == BRICK_STATS) {
        // exact integer totals; one atomic per lane that has something to add
        if (st_entries) atomicAdd(&a.stats[0], st_entries);
        if (st_max) atomicMax(&a.stats[1], (unsigned long long)st_max);
        if (st_inside) atomicAdd(&a.stats[2], st_inside);
    }
}

// Largest tile (brick + halo population) and largest own population over all bricks -> sizes the
// dynamic LDS of k_brick.  out[0] = max tile, out[1] = max own.
template <class Shape>
__global__ void k_brick_tile_max(BrickGrid bg, int Mx, int My, int Mz, int px, int py, int pz,
                                 const int *__restrict__ start, int *__restrict__ out) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    int total = 0, own = 0;
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
                for (int tx = 0; tx < Shape::TX; tx++) {
                    const int rx = ox0 - 1 + tx;
                    int gx = rx;
                    if (gx > ox1) continue;
                    if (gx < 0) { if (!px) continue; gx += Mx; } else if (gx >= Mx) { if (!px) continue; gx -= Mx; }
                    const int c = gx + Mx * (gy + My * gz);
                    const int pop = start[c + 1] - start[c];
                    total += pop;
                    if (rx >= ox0 && rx < ox1 && ry >= oy0 && ry < oy1 && rz >= oz0 && rz < oz1) own += pop;
                }
            }
    }
#pragma unroll
    for (int off = 1; off < WAVE; off <<= 1) {
        total = max(total, __shfl_xor(total, off));
        own = max(own, __shfl_xor(own, off));
    }
    if ((threadIdx.x & (WAVE - 1)) == 0 && total > 0) {
        atomicMax(&out[0], total);
        atomicMax(&out[1], own);
    }
}

}  // namespace emdee
