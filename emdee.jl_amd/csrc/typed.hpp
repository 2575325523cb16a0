// typed.hpp -- LDS-tiled neighbour build and force kernels for boxes of TWO species (a binary mixture, BASELINE configs[4]).
//
// Why.  The general-species kernels of brick.hpp mix the Lorentz-Berthelot parameters PER PAIR: the neighbour's LJAtom
// comes out of the 32-byte tile record, is converted and combined (2 cvt, add, 2 mul) before sigma_ij^2 / r^2 and
// 4 eps_ij can enter the pair function: 39 VALU instructions per pair step against 31 in the single-species kernels,
// at the same issue efficiency (round 3: 2.2 ps against 1.79 ps per listed entry).  The reference encodes the mixing rule
// per atom (src/lennard_jones.jl:13-18,29-30), so the pair constants depend on the two SPECIES only -- with two species
// there are four of them.  Here the box is sorted by (cell, species), the LDS tile of a brick is staged species-major
// (all tile cells' atoms of species 0, then of species 1: inside one species the cells keep the tile order, so the
// three cells of a candidate row stay contiguous), the build kernel walks 9 candidate rows per species and therefore
// emits a neighbour row as TWO SEGMENTS -- neighbours of species 0, then, from the next 8 G-entry block boundary on,
// neighbours of species 1 -- and the force kernel runs the pair loop once per segment with (sigma_ij^2, 4 eps_ij) of
// (species_i, segment) in registers: no parameter gather, no conversion, no mixing per pair, and coordinate planes
// (24 B per record) instead of 32-byte records in LDS.  The candidate rows are half as long as the untyped ones, so
// the 16-bit-field round-robin build of the single-species boxes (ALG 13 of brick.hpp) applies at rc = 3.5 sigma too.
//
// Same arithmetic per pair as the general-species kernels (sigma_ij^2 and 4 eps_ij are formed on the host with the very
// operations the pair loop used, lj_pair.hpp lj_force_over_r2 / lj_interaction_pair evaluate them): results differ only
// in the order of summation.  Restates what brick.hpp restates: src/lennard_jones.jl:25-42, src/nonbonded.jl:136-145,
// src/cells.jl:224-297.
#pragma once

#include "brick.hpp"

namespace emdee {

constexpr int TNT = 2;   // species of a typed box

// planes of a typed tile: records per plane (compile time: the three reads of a neighbour share one address register).
// 512-thread workgroups keep the single-species pitch (three workgroups per CU); 1024-thread ones -- long cutoffs: the
// rc = 3.5 sigma tile holds ~4400 records -- take one workgroup per CU and 4608 records.
// (round 5: bricks of 2 x 2 x 2 cells -- 64 tile cells, 2916 records at rc = 3.5 sigma and rho* = 0.8 before the melt's fluctuations; 3104 slots are what half a
// CU's LDS holds in fp64 next to the tables -- in 512-thread workgroups, TWO per CU:
// with one 1024-thread workgroup per CU the vector units idled 30 % of the kernel -- nothing computes while a tile is staged
// or a workgroup is replaced, profiles/r05/valu_f64_mix_rc3.5.txt)
template <class Shape, int THREADS>
constexpr int typed_slots() { return THREADS >= 1024 ? 4608 : (Shape::NOC == 8 ? 3104 : SOA_SLOTS); }
template <typename real, class Shape, int THREADS>
constexpr int typed_pitch() { return typed_slots<Shape, THREADS>() + ((sizeof(real) == 8 && EMDEE_SOA_PAD) ? 1 : 0); }
// index blocks of a row (= of its species-0 segment) fetched one atom ahead: two of 32 entries where rows are short, four where
// the workgroup is a long-row one (rc = 3.5 sigma: ~92 neighbours per species; NOC = own cells of the brick shape)
constexpr int typed_prefetch_blocks(int G, int THREADS, int NOC = 16) { return (EPL * G) >= 128 ? 1 : ((G == 4 && (THREADS >= 1024 || NOC == 8)) ? 4 : 2); }
// entries the build's LDS row buffer holds per atom: ONE segment (the two species are emitted and flushed one after the other)
constexpr int typed_seg_cap(int stride, int GL) { return ((stride / 2) + EPL * GL - 1) / (EPL * GL) * (EPL * GL); }

template <class Shape, int THREADS>
struct TypedTables {
    static constexpr int NTC = Shape::NTC, NOC = Shape::NOC, NTT = TNT * NTC, NOT = TNT * NOC, NWAVES = THREADS / WAVE;
    int *off;      // [NTT+1] tile-local first slot of (species, tile cell), species-major
    int *gbeg;     // [NTT]   global (cell-order) first slot of that block
    int *shift;    // [NTC]   periodic image of the tile cell: 2 bits per dimension (0:-1, 1:0, 2:+1)
    int *own;      // [NOT+1] prefix of the own populations, (species, own cell)
    int *wtot;     // [NWAVES]
    int2 *oinfo;   // [own_cap] per own atom (filled by the kernel)
    int *ocnt;     // [own_cap] ... and its row lengths n0 | n1 << 16 (force kernels)
    __host__ __device__ static constexpr size_t fixed_ints() { return (NTT + 4) + NTT + NTC + (NOT + 4) + ((NWAVES + 1) & ~1); }
    __host__ __device__ static size_t bytes(int own_cap) { return ((fixed_ints() + 3 * (size_t)own_cap) * 4 + 15) & ~(size_t)15; }
    // off | gbeg | shift | own are contiguous: that image (+ tile_n, n_own) is what k_typed_tables stores per brick
    __host__ __device__ static constexpr int image_ints() { return (NTT + 4) + NTT + NTC + (NOT + 4); }
    __host__ __device__ static constexpr int row_ints() { return (image_ints() + 2 + 3) & ~3; }
    __device__ __forceinline__ void carve(unsigned char *base) {
        off = reinterpret_cast<int *>(base);
        gbeg = off + (NTT + 4);
        shift = gbeg + NTT;
        own = shift + NTC;
        wtot = own + (NOT + 4);
        oinfo = reinterpret_cast<int2 *>(wtot + ((NWAVES + 1) & ~1));
    }
    __device__ __forceinline__ void carve_counts(int own_cap) { ocnt = reinterpret_cast<int *>(oinfo + own_cap); }
};

template <typename real, class Shape, int THREADS>
static inline size_t typed_force_lds_bytes(int own_cap) {
    return (((size_t)3 * typed_pitch<real, Shape, THREADS>() * sizeof(real) + 15) & ~(size_t)15) + TypedTables<Shape, THREADS>::bytes(own_cap) + 256;   // (+ the four pair-constant records)
}
template <class Shape, int THREADS>
static inline size_t typed_build_lds_bytes(int tile_cap, int own_cap, int stride, int G, int GL = 4) {
    return (size_t)tile_cap * 16 + TypedTables<Shape, THREADS>::bytes(own_cap) + (size_t)(THREADS / G) * typed_seg_cap(stride, GL) * 2 +
           (((size_t)(Shape::NOC * 4 + 1) * 9 * TNT * 4 + 15) & ~(size_t)15);   // (row table: one packed word per (own cell, x quarter, species, row))
}

// tables of this block's brick; false when the block has nothing to do.  Contains block barriers.
template <typename real, class Shape, int THREADS, bool COMPUTE = false>
__device__ __forceinline__ bool typed_setup(const BrickArgs<real> &a, const TypedTables<Shape, THREADS> &T, int &bxi, int &byi,
                                            int &bzi, int &tile_n, int &n_own) {
    constexpr int BX = Shape::BX, BY = Shape::BY, BZ = Shape::BZ, TX = Shape::TX, TY = Shape::TY, NTC = Shape::NTC, NOC = Shape::NOC;
    constexpr int NTT = TNT * NTC, NOT = TNT * NOC;
    static_assert(NTT <= THREADS, "typed tables: one thread per (species, tile cell)");
    const int tid = threadIdx.x, lane = tid & (WAVE - 1), wv = tid / WAVE;
    const int Mx = a.g.M[0], My = a.g.M[1], Mz = a.g.M[2];
    if (!brick_of_block(a, bxi, byi, bzi)) return false;
    if (!COMPUTE && a.btab != nullptr) {
        constexpr int ROW = TypedTables<Shape, THREADS>::row_ints(), IMG = TypedTables<Shape, THREADS>::image_ints();
        const int *row = a.btab + (size_t)(bxi + a.bg.nb[0] * (byi + a.bg.nb[1] * bzi)) * ROW;
        for (int i = tid; i < ROW / 4; i += THREADS)
            reinterpret_cast<uint4 *>(T.off)[i] = reinterpret_cast<const uint4 *>(row)[i];
        __syncthreads();
        tile_n = T.off[IMG];
        n_own = T.off[IMG + 1];
        if (tile_n > a.tile_cap || n_own > a.own_cap) {   // (a kept plan the populations outgrew: skipped and reported)
            if (tid == 0) atomicMax(&a.flags[2], max(tile_n, n_own));
            return false;
        }
        return n_own > 0;
    }
    int my_cnt = 0;
    if (tid < NTT) {
        const int t = tid / NTC, tc = tid % NTC;
        const int tx = tc % TX, ty = (tc / TX) % TY, tz = tc / (TX * TY);
        int gx = bxi * BX - 1 + tx, gy = byi * BY - 1 + ty, gz = bzi * BZ - 1 + tz;
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
        int gb = 0, packed = 0;
        if (valid) {
            const size_t c = ((size_t)(gx + Mx * (gy + My * gz)) * TNT + t) * a.tdig;
            gb = a.tstart[c];
            my_cnt = a.tstart[c + a.tdig] - gb;
            if (COMPUTE && a.bsub != nullptr) {          // x sub-bins: the three inner boundaries of the block (brick.hpp sub_below)
                const int *fs = a.tstart + c;
                packed = (fs[1] - gb) | ((fs[2] - gb) << 10) | ((fs[3] - gb) << 20);
            }
        }
        T.gbeg[tid] = gb;
        if (t == 0) T.shift[tc] = sh;
        if (COMPUTE && a.bsub != nullptr) a.bsub[(size_t)(bxi + a.bg.nb[0] * (byi + a.bg.nb[1] * bzi)) * NTT + tid] = packed;
    }
    // Decomposed runs: an own (species, cell) block that holds nothing but ghosts takes no part in the own-atom loops (as in
    // brick.hpp): bit 8 + species of the cell's shift word, set once the plain words are in place
    int ghosts_only = 0;
    if (COMPUTE && a.any_ghosts && tid < NTT && my_cnt > 0) {
        const int tc = tid % NTC, tx = tc % TX, ty = (tc / TX) % TY, tz = tc / (TX * TY);
        if (tx >= 1 && tx <= BX && ty >= 1 && ty <= BY && tz >= 1 && tz <= BZ) {
            int owned = 0;
            const int gb = T.gbeg[tid];
            for (int k = 0; k < my_cnt; k++) owned |= (a.perm[gb + k] < a.n_owned) ? 1 : 0;
            ghosts_only = owned ? 0 : 1;
        }
    }
    {   // exclusive scan of my_cnt over the first NTT threads; tile slots start at 1 (slot 0 is the sentinel record)
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
        if (tid < NTT) T.off[tid] = 1 + woff + inc - my_cnt;
        if (tid == NTT - 1) T.off[NTT] = 1 + woff + inc;
        if (ghosts_only) atomicOr(&T.shift[tid % NTC], 256 << (tid / NTC));   // (the shift words were stored before the barrier above)
    }
    __syncthreads();
    tile_n = T.off[NTT];
    if (tid == 0) {
        int acc = 0;
        for (int q = 0; q < NOT; q++) {
            const int t = q / NOC, oc = q % NOC;
            const int ox = oc % BX, oy = (oc / BX) % BY, oz = oc / (BX * BY);
            const int tc = t * NTC + (ox + 1) + TX * ((oy + 1) + TY * (oz + 1));
            T.own[q] = acc;
            const bool mine = (bxi * BX + ox < Mx) && (byi * BY + oy < My) && (bzi * BZ + oz < Mz) &&
                              !(T.shift[tc - t * NTC] & (256 << t));
            acc += mine ? (T.off[tc + 1] - T.off[tc]) : 0;
        }
        T.own[NOT] = acc;
    }
    __syncthreads();
    n_own = T.own[NOT];
    if (tile_n > a.tile_cap || n_own > a.own_cap) {
        if (tid == 0) atomicMax(&a.flags[2], max(tile_n, n_own));
        return false;
    }
    return n_own > 0;
}

// own atom o -> (species, own cell) index q, tile slot, cell-order slot
template <class Shape, int THREADS>
__device__ __forceinline__ int typed_locate(const TypedTables<Shape, THREADS> &T, int o, int &ti, int &p, int *tc_out = nullptr) {
    constexpr int NOC = Shape::NOC, NTC = Shape::NTC;
    int q = 0;
#pragma unroll
    for (int k = 1; k < TNT * NOC; k++) q += (T.own[k] <= o) ? 1 : 0;
    const int t = q / NOC, oc = q % NOC;
    const int ox = oc % Shape::BX, oy = (oc / Shape::BX) % Shape::BY, oz = oc / (Shape::BX * Shape::BY);
    const int tc = t * NTC + (ox + 1) + Shape::TX * ((oy + 1) + Shape::TY * (oz + 1));
    const int kk = o - T.own[q];
    ti = T.off[tc] + kk;
    p = T.gbeg[tc] + kk;
    if (tc_out) *tc_out = tc;
    return q;
}

// f(slot, typed tile cell) once for every tile slot; half-waves take whole (species, tile row) runs
template <class Shape, int THREADS, class F>
__device__ __forceinline__ void typed_for_each_slot(const TypedTables<Shape, THREADS> &T, F &&f) {
    constexpr int NW = THREADS / STAGE_LANES, NYZ = Shape::TY * Shape::TZ, NROWS = NYZ * TNT, TX = Shape::TX;
    const int worker = threadIdx.x / STAGE_LANES, l = threadIdx.x % STAGE_LANES;
    for (int row = worker; row < NROWS; row += NW) {
        const int c0 = (row / NYZ) * Shape::NTC + (row % NYZ) * TX;
        const int ty = (row % NYZ) % Shape::TY, tz = (row % NYZ) / Shape::TY;   // (per row: the cell-relative staging indexes by them)
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

// once per rebuild: the tables of every brick, stored as the LDS image the other kernels copy in.  With a.stats != NULL
// also the population maxima of the plan check, per species: [2] = most atoms of ONE species in three consecutive cells.
template <typename real, class Shape, int THREADS>
__global__ __launch_bounds__(THREADS) void k_typed_tables(BrickArgs<real> a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char s_dyn[];
    TypedTables<Shape, THREADS> T;
    T.carve(s_dyn);
    const int lb = (blockIdx.x % NXCD) * a.bg.per_xcd + blockIdx.x / NXCD;
    if (lb >= a.bg.nbricks) return;
    int bxi = 0, byi = 0, bzi = 0, tile_n = 0, n_own = 0;
    typed_setup<real, Shape, THREADS, true>(a, T, bxi, byi, bzi, tile_n, n_own);
    constexpr int ROW = TypedTables<Shape, THREADS>::row_ints(), IMG = TypedTables<Shape, THREADS>::image_ints();
    int *row = a.btab + (size_t)(bxi + a.bg.nb[0] * (byi + a.bg.nb[1] * bzi)) * ROW;
    for (int i = threadIdx.x; i < IMG; i += THREADS) row[i] = T.off[i];
    if (threadIdx.x == 0) { row[IMG] = tile_n; row[IMG + 1] = n_own; }
}

// most atoms of one species in three consecutive cells of a tile row, over all bricks (-> out[0]); what the 16-bit hit
// fields of the typed build can take is 16 G
template <class Shape>
__global__ void k_typed_span_max(BrickGrid bg, int Mx, int My, int Mz, int px, int py, int pz, const int *__restrict__ tstart,
                                 int *__restrict__ out, int tdig = 1) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    int span3 = 0;
    if (b < bg.nbricks) {
        const int bxi = b % bg.nb[0], byi = (b / bg.nb[0]) % bg.nb[1], bzi = b / (bg.nb[0] * bg.nb[1]);
        const int ox0 = bxi * Shape::BX, oy0 = byi * Shape::BY, oz0 = bzi * Shape::BZ;
        const int ox1 = min(ox0 + Shape::BX, Mx), oy1 = min(oy0 + Shape::BY, My), oz1 = min(oz0 + Shape::BZ, Mz);
        for (int tz = 0; tz < Shape::TZ; tz++)
            for (int ty = 0; ty < Shape::TY; ty++) {
                int gy = oy0 - 1 + ty, gz = oz0 - 1 + tz;
                if (gy > oy1 || gz > oz1) continue;
                if (gy < 0) { if (!py) continue; gy += My; } else if (gy >= My) { if (!py) continue; gy -= My; }
                if (gz < 0) { if (!pz) continue; gz += Mz; } else if (gz >= Mz) { if (!pz) continue; gz -= Mz; }
                int p1[TNT] = {0, 0}, p2[TNT] = {0, 0};
                for (int tx = 0; tx < Shape::TX; tx++) {
                    int gx = ox0 - 1 + tx;
                    bool valid = gx <= ox1;
                    if (gx < 0) { if (!px) valid = false; gx += Mx; } else if (gx >= Mx) { if (!px) valid = false; gx -= Mx; }
                    for (int t = 0; t < TNT; t++) {
                        int pop = 0;
                        if (valid) {
                            const size_t c = ((size_t)(gx + Mx * (gy + My * gz)) * TNT + t) * tdig;
                            pop = tstart[c + tdig] - tstart[c];
                        }
                        span3 = max(span3, pop + p1[t] + p2[t]);
                        p2[t] = p1[t]; p1[t] = pop;
                    }
                }
            }
    }
#pragma unroll
    for (int off = 1; off < WAVE; off <<= 1) span3 = max(span3, __shfl_xor(span3, off));
    if ((threadIdx.x & (WAVE - 1)) == 0 && span3 > 0) atomicMax(&out[0], span3);
}

// ------------------------------------------------------------------------------------ build
// ALG 13 of brick.hpp (round-robin candidates, wave-uniform trip counts from one reduction per species, the listed bit
// through one v_alignbit_b32, hits in plain order and the lane-major layout produced at the flush) over 9 candidate rows
// per species.  Row of atom p: entries [0, n0) = neighbours of species 0, [S1, S1 + n1) = neighbours of species 1 with
// S1 = n0 rounded up to a whole block of 8 GL entries; cnt[p] = n0 | n1 << 16.
template <typename real, class Shape, int THREADS, int G, int GL>
__global__ __launch_bounds__(THREADS, (THREADS <= 512 && Shape::NOC != 8 ? 5 : 4)) void k_typed_build(BrickArgs<real> a) {   // (2 x 2 x 2 bricks: two 512-thread workgroups per CU = 4 waves per SIMD; short-row boxes, EMDEE_TYPED_ALL: 5 -- at 6 the build spilled 12 registers per lane once it took the x quarters)
    constexpr int BX = Shape::BX, BY = Shape::BY, TX = Shape::TX, TY = Shape::TY, NTC = Shape::NTC, NOC = Shape::NOC;
    constexpr int NGROUPS = (THREADS / WAVE) * (WAVE / G);
    static_assert(G == 8, "typed build: 8 lanes per atom");
    constexpr int LOG2G = 3, NROWS = 9, NR2 = NROWS * TNT, WPT = (NROWS + 1) / 2, NWORDS = WPT * TNT;   // 5 words per species
    constexpr bool BAND = sizeof(real) == 8;
    extern __shared__ __attribute__((aligned(16))) unsigned char s_dyn[];
    float4 *tile = reinterpret_cast<float4 *>(s_dyn);   // {x, y, z relative to the brick origin, cell-order slot}
    TypedTables<Shape, THREADS> T;
    T.carve(s_dyn + (size_t)a.tile_cap * 16);
    int bxi, byi, bzi, tile_n, n_own;
    if (!typed_setup<real, Shape, THREADS>(a, T, bxi, byi, bzi, tile_n, n_own)) return;
    const int tid = threadIdx.x, lane = tid & (WAVE - 1);
    // x sub-bins (round 5; as brick.hpp's builds): a (cell, species) block is sorted by quarter along x, and an atom of quarter s
    // takes the quarters >= s + K of the left cell of a candidate row, the middle cell and the quarters <= s - K of the right
    // one -- 9 of 12 quarters where cells are just wider than r_list.  The boundaries of the tile's blocks (k_typed_tables
    // wrote them) borrow the row buffers until the row table is in place.
    constexpr int NTT = TNT * NTC;
    const int NSUB = (a.nsub == 4 && a.bsub != nullptr) ? 4 : 1;
    unsigned char *after = s_dyn + (size_t)a.tile_cap * 16 + TypedTables<Shape, THREADS>::bytes(a.own_cap);
    int *s_sub = reinterpret_cast<int *>(after);
    if (NSUB > 1) {
        const int *bs = a.bsub + (size_t)(bxi + a.bg.nb[0] * (byi + a.bg.nb[1] * bzi)) * NTT;
        for (int i = tid; i < NTT; i += THREADS) s_sub[i] = bs[i];
    }

    real org[3] = {0, 0, 0};
    if (sizeof(real) == 8) {
        org[0] = a.g.lo[0] + (real)(bxi * BX) * (a.g.len[0] / (real)a.g.M[0]);
        org[1] = a.g.lo[1] + (real)(byi * BY) * (a.g.len[1] / (real)a.g.M[1]);
        org[2] = a.g.lo[2] + (real)(bzi * Shape::BZ) * (a.g.len[2] / (real)a.g.M[2]);
    }
    __shared__ float s_relc[sizeof(real) == 4 ? Shape::TX + Shape::TY + Shape::TZ : 1];   // cell-relative records (brick.hpp rel_cell_const)
    if (sizeof(real) == 4 && a.rel) rel_fill_consts<real, Shape>(a, bxi, byi, bzi, s_relc);
    if (NSUB > 1 || (sizeof(real) == 4 && a.rel)) __syncthreads();
    // quarter of the own atom at tile slot ti of typed tile cell tc
    auto own_sub = [&](int tc, int ti) {
        if (NSUB == 1) return 0;
        const int kk = ti - T.off[tc], pk = s_sub[tc];
        return (kk >= (pk & 1023) ? 1 : 0) + (kk >= ((pk >> 10) & 1023) ? 1 : 0) + (kk >= ((pk >> 20) & 1023) ? 1 : 0);
    };
    int own_p[OWN_REGS], own_ti[OWN_REGS], own_key[OWN_REGS], own_q[OWN_REGS];
#pragma unroll
    for (int k = 0; k < OWN_REGS; k++) {
        const int o = tid + k * THREADS;
        own_p[k] = own_ti[k] = 0; own_key[k] = 0; own_q[k] = 0;
        if (o < n_own) {
            int tc;
            own_q[k] = typed_locate(T, o, own_ti[k], own_p[k], &tc);
            own_key[k] = (a.perm[own_p[k]] < a.n_owned ? 1 : 0) | (own_sub(tc, own_ti[k]) << 1);
        }
    }
    typed_for_each_slot(T, [&](int s, int tc, int tx, int ty, int tz) {
        const int gp = T.gbeg[tc] + (s - T.off[tc]);
        const int sh = T.shift[tc % NTC];
        const Rec<real> r = a.rec[gp];
        float4 q;
        if (sizeof(real) == 4 && a.rel) {                     // cell-relative records (brick.hpp rel_tile)
            q.x = (float)rel_tile(r.x, s_relc[tx]);
            q.y = (float)rel_tile(r.y, s_relc[TX + ty]);
            q.z = (float)rel_tile(r.z, s_relc[TX + TY + tz]);
        } else {
            q.x = (float)((r.x + (real)((sh & 3) - 1) * a.g.len[0]) - org[0]);
            q.y = (float)((r.y + (real)(((sh >> 2) & 3) - 1) * a.g.len[1]) - org[1]);
            q.z = (float)((r.z + (real)(((sh >> 4) & 3) - 1) * a.g.len[2]) - org[2]);
        }
        q.w = __int_as_float(gp);
        if (EMDEE_BOUND(BS_TYPED_TILE, s, a.tile_cap)) tile[s] = q;
    });
#pragma unroll
    for (int k = 0; k < OWN_REGS; k++) {
        const int o = tid + k * THREADS;
        // (own_key: bit 0 = owned, bits 1-2 = x quarter)
        if (o < n_own && EMDEE_BOUND(BS_TYPED_OWN, o, a.own_cap)) T.oinfo[o] = make_int2(own_p[k], (own_q[k] << 20) | (own_key[k] << 16) | own_ti[k]);
    }
    for (int o = tid + OWN_REGS * THREADS; o < n_own; o += THREADS) {
        int ti, p, tc;
        const int q = typed_locate(T, o, ti, p, &tc);
        if (EMDEE_BOUND(BS_TYPED_OWN, o, a.own_cap)) T.oinfo[o] = make_int2(p, (q << 20) | (own_sub(tc, ti) << 17) | ((a.perm[p] < a.n_owned ? 1 : 0) << 16) | ti);
    }
    // candidate rows per own cell (and x quarter): for each neighbour species the 9 tile rows of 3 cells around it, packed as
    // first tile slot | slots << 16 (a tile holds < 2^16 slots); the last entry is empty (atoms that own no row)
    unsigned *rtab = reinterpret_cast<unsigned *>(after + (size_t)NGROUPS * typed_seg_cap(a.stride, GL) * 2);
    for (int i = tid; i < (NOC * NSUB + 1) * NR2; i += THREADS) {
        const int ocs = i / NR2, tr = i % NR2, t = tr / NROWS, r = tr % NROWS;
        unsigned v = 0;
        if (ocs < NOC * NSUB) {
            const int oc = ocs / NSUB, sb = ocs - oc * NSUB;
            const int ox = oc % BX, oy = (oc / BX) % BY, oz = oc / (BX * BY);
            const int tcr = t * NTC + ox + TX * ((oy + r % 3) + TY * (oz + r / 3));   // cell x-1 of tile row (dy, dz) of species t
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
    constexpr int BLKL = EPL * GL;                           // entries per lane-major block of the force kernels' rows
    const int segcap = typed_seg_cap(a.stride, GL);          // the row buffer holds ONE segment (round 5: half the LDS of a whole row)
    unsigned short *rowbuf = reinterpret_cast<unsigned short *>(after) + (size_t)gid * segcap;
    const unsigned ustride = (unsigned)a.stride;
    const uint4 fill = make_uint4(0, 0, 0, 0);               // sentinel slot 0
    float nrl2 = -rl2, margin_v = a.margin;
    asm volatile("" : "+v"(nrl2), "+v"(margin_v));
    const int kshift = a.idx_shift + LOG2G;                  // bit k of a field is slot cb + k G
    for (int ob = 0; ob < n_own; ob += NGROUPS) {            // wave-uniform trip count
        const int o = ob + gid;
        const bool have = o < n_own;
        const int2 info = have ? T.oinfo[o] : make_int2(0, 0);
        const int ti = info.y & 0xffff, p = info.x, q_own = info.y >> 20, oc = q_own % NOC, t_own = q_own / NOC;
        const bool act = have && ((info.y >> 16) & 1) != 0;   // ghosts own no row
        const float4 qi = tile[ti];
        unsigned short *row = a.nbr + (size_t)p * a.stride;
        const unsigned *rt = rtab + (act ? oc * NSUB + ((info.y >> 17) & 3) : NOC * NSUB) * NR2;
        // wave-uniform trip counts of the 18 rows: per species, lane gl of every group holds the chunk length of row gl
        // (row 8 apart), three max steps combine the groups of the wavefront
        int trips_of[NR2];
        auto rows_max = [](int v) {
            auto q = __builtin_amdgcn_permlane16_swap((unsigned)v, (unsigned)v, false, false);
            v = max((int)q[0], (int)q[1]);
            q = __builtin_amdgcn_permlane32_swap((unsigned)v, (unsigned)v, false, false);
            return max((int)q[0], (int)q[1]);
        };
#pragma unroll
        for (int t = 0; t < TNT; t++) {
            int cv = (int)(((rt[t * NROWS + min(gl, NROWS - 1)] >> 16) + G - 1) / (unsigned)G);
            int c8 = (int)(((rt[t * NROWS + NROWS - 1] >> 16) + G - 1) / (unsigned)G);
            cv = max(cv, __builtin_amdgcn_update_dpp(0, cv, 0x128 /* row_ror:8 */, 0xf, 0xf, true));
            c8 = max(c8, __builtin_amdgcn_update_dpp(0, c8, 0x128, 0xf, 0xf, true));
            cv = rows_max(cv);
            c8 = rows_max(c8);
#pragma unroll
            for (int r = 0; r < NROWS; r++)
                trips_of[t * NROWS + r] = (r == NROWS - 1) ? __builtin_amdgcn_readlane(c8, 0) : __builtin_amdgcn_readlane(cv, r);
        }
        {   // (a kept plan: three cells may have come to hold more atoms of one species than the 16-bit fields take -- report it,
            // the host plans afresh; nothing is written for this brick)
            int tmax = 0;
#pragma unroll
            for (int tr = 0; tr < NR2; tr++) tmax = max(tmax, trips_of[tr]);
            if (tmax > 16) {
                if (lane == 0) atomicMax(&a.flags[2], 0x7ffffff0);
                return;
            }
        }
        unsigned word[NWORDS];
#pragma unroll
        for (int w = 0; w < NWORDS; w++) word[w] = 0;
        unsigned rv_next = rt[0];
#pragma unroll
        for (int tr = 0; tr < NR2; tr++) {
            const int t = tr / NROWS, r = tr % NROWS;
            const int c0 = (int)(rv_next & 0xffffu), span = (int)(rv_next >> 16);
            if (tr + 1 < NR2) rv_next = rt[tr + 1];
            const int lim = (int)((unsigned)(span - gl + G - 1) >> LOG2G);   // my candidates: slots cb + k G, k < lim (may be <= 0)
            const int cb = c0 + gl;
            const int trips = trips_of[tr];                      // <= 16 (host check: three cells hold <= 16 G atoms of one species)
            unsigned bits = 0;
            const float4 *cand = tile + cb;
            auto dist = [&](const float4 &q) {
                const float dx = qi.x - q.x, dyy = qi.y - q.y, dzz = qi.z - q.z;
                float tt = __builtin_fmaf(dx, dx, nrl2);
                tt = __builtin_fmaf(dyy, dyy, tt);
                return __builtin_fmaf(dzz, dzz, tt);
            };
            // rounding band (fp64 boxes): decided with the exact fp64 records
            auto exact = [&](float &tt, int gpj, int k) {
                if (__builtin_fabsf(tt) <= margin_v && k < lim) {
                    int kq = k;
                    asm volatile("" : "+s"(kq));
                    const int c = cb + kq * G;
                    int occ = oc;
                    asm volatile("" : "+v"(occ));
                    const int cell = occ % BX + TX * (((occ / BX) % BY + r % 3) + TY * (occ / (BX * BY) + r / 3));   // untyped tile cell x-1
                    const int tcr = t * NTC + cell;
                    const int k3 = (c >= T.off[tcr + 1] ? 1 : 0) + (c >= T.off[tcr + 2] ? 1 : 0);
                    const int sh = T.shift[cell + k3];
                    const Rec<real> ri = a.rec[p], rj = a.rec[gpj];
                    const real ex = ri.x - (rj.x + (real)((sh & 3) - 1) * a.g.len[0]);
                    const real ey = ri.y - (rj.y + (real)(((sh >> 2) & 3) - 1) * a.g.len[1]);
                    const real ez = ri.z - (rj.z + (real)(((sh >> 4) & 3) - 1) * a.g.len[2]);
                    tt = (ex * ex + ey * ey + ez * ez < a.rlist2) ? -1.f : 1.f;
                }
            };
            auto shift_in = [&](float tt) { bits = __builtin_amdgcn_alignbit(bits, __float_as_uint(tt), 31); };   // bits = 2 bits + (t < 0)
#pragma clang loop unroll(disable) vectorize(disable) interleave(disable)
            for (int k = 0; k + 1 < trips; k += 2) {             // (reads past my share / the tile: harmless)
                float4 q[2];
                float tt[2];
                q[0] = cand[k * G]; q[1] = cand[(k + 1) * G];
                if constexpr (!BAND) { asm volatile("" : : "v"(q[0].w)); asm volatile("" : : "v"(q[1].w)); }
                tt[0] = dist(q[0]); tt[1] = dist(q[1]);
                if constexpr (BAND) {
                    const float tm = __builtin_fminf(__builtin_fabsf(tt[0]), __builtin_fabsf(tt[1]));
                    if (__builtin_expect(__builtin_amdgcn_ballot_w64(tm <= margin_v) != 0, 0)) {
                        exact(tt[0], __float_as_int(q[0].w), k);
                        exact(tt[1], __float_as_int(q[1].w), k + 1);
                    }
                }
                shift_in(tt[0]); shift_in(tt[1]);
            }
            if (trips & 1) {
                const float4 q = cand[(trips - 1) * G];
                if constexpr (!BAND) asm volatile("" : : "v"(q.w));
                float tt = dist(q);
                if constexpr (BAND) {
                    if (__builtin_expect(__builtin_amdgcn_ballot_w64(__builtin_fabsf(tt) <= margin_v) != 0, 0))
                        exact(tt, __float_as_int(q.w), trips - 1);
                }
                shift_in(tt);
            }
            // candidate k sits at bit trips-1-k: reverse, drop what lies past my share
            bits = __builtin_amdgcn_ubfe(__builtin_bitreverse32(bits), (unsigned)(32 - trips), (unsigned)max(lim, 0));
            if (r == 4 && t == t_own) {                          // the atom itself (its cell is the middle one of row 4 of its species)
                const int d = ti - c0;
                if (d >= 0 && (d & (G - 1)) == gl) bits &= ~(1u << (d >> LOG2G));
            }
            // rows (dy, dz) and (-dy, -dz) share a word: (0,8) (1,7) (2,6) (3,5) (4) -- their hits add up to nearly the same number
            // for every atom, and the emission loop of a word runs as long as the busiest lane of the wavefront (brick.hpp)
            const int w = t * WPT + (r <= 4 ? r : 8 - r);
            if (r > 4) word[w] |= bits << 16;
            else word[w] = bits;
        }
        // ---- phase 2: per species, prefix over the lanes of the group, then every lane emits its hits ----------------
        int mine0 = 0, mine1 = 0;
#pragma unroll
        for (int w = 0; w < WPT; w++) { mine0 += __popc(word[w]); mine1 += __popc(word[WPT + w]); }
        auto group_prefix = [&](int v) {
            int t = __builtin_amdgcn_update_dpp(0, v, DPP_ROW_SHR1, 0xf, 0xf, true);
            v += gl >= 1 ? t : 0;
            t = __builtin_amdgcn_update_dpp(0, v, DPP_ROW_SHR2, 0xf, 0xf, true);
            v += gl >= 2 ? t : 0;
            t = __builtin_amdgcn_update_dpp(0, v, DPP_ROW_SHR4, 0xf, 0xf, true);
            v += gl >= 4 ? t : 0;
            return v;
        };
        const int incl0 = group_prefix(mine0), incl1 = group_prefix(mine1);
        const int total0 = __shfl(incl0, lane | (G - 1)), total1 = __shfl(incl1, lane | (G - 1));
        const int S1 = (total0 + BLKL - 1) / BLKL * BLKL;        // species-1 segment: from the next block boundary on
        unsigned short *const ep_last = rowbuf + (unsigned)(segcap - 1);
        // One segment at a time through the row buffer: fill with sentinels, emit the species' hits, flush its blocks of the
        // row -- [0, S1) for species 0, [S1, stride) for species 1 (whatever lies behind the last entry is sentinels: a lane
        // of the force kernel walks as far as the longest row of its wavefront)
#pragma unroll
        for (int t = 0; t < TNT; t++) {
            for (int c = gl * EPL; c < segcap; c += G * EPL) *reinterpret_cast<uint4 *>(rowbuf + c) = fill;
            unsigned short *ep = rowbuf + (unsigned)(t == 0 ? incl0 - mine0 : incl1 - mine1);
#pragma unroll
            for (int w = 0; w < WPT; w++) {
                unsigned W = word[t * WPT + w];
                // field A = row w of this species (bits 0..15), field B = the opposite row 8 - w (bits 16..31)
                int cA = ((int)(rt[t * NROWS + w] & 0xffffu) + gl) << a.idx_shift;
                int cB = (w < 4) ? (((int)(rt[t * NROWS + 8 - w] & 0xffffu) + gl - 16 * G) << a.idx_shift) : 0;
                asm volatile("" : "+v"(cA), "+v"(cB));
                while (W) {
                    const int k = __ffs((int)W) - 1;
                    W &= W - 1;
                    asm volatile("" : "+v"(W));
                    *(ep < ep_last ? ep : ep_last) = (unsigned short)((k << kshift) + (k >= 16 ? cB : cA));
                    ep++;
                }
            }
            if (have && EMDEE_BOUND(BS_TYPED_ROW, p, a.n)) {
                const int lo = t == 0 ? 0 : S1, hi = t == 0 ? min(S1, a.stride) : a.stride;
                // (left to itself the compiler interleaves two trips of this loop behind a run-time alias check and splits each
                // 16-byte store into four: 130 instructions per atom where 40 do -- the flush was 0.32 ms of the build for that)
                EMDEE_PLAIN_LOOP
                for (int c = lo + gl * EPL; c < hi; c += G * EPL) {
                    const int cs = c - lo;                    // position inside the segment
                    uint4 q = fill;
                    if (cs < segcap) {
                        const unsigned short *src = rowbuf + (cs / BLKL) * BLKL + (cs % BLKL) / EPL;   // entries src[GL t], t = 0..7
                        q.x = (unsigned)src[0 * GL] | ((unsigned)src[1 * GL] << 16);
                        q.y = (unsigned)src[2 * GL] | ((unsigned)src[3 * GL] << 16);
                        q.z = (unsigned)src[4 * GL] | ((unsigned)src[5 * GL] << 16);
                        q.w = (unsigned)src[6 * GL] | ((unsigned)src[7 * GL] << 16);
                    }
                    *reinterpret_cast<uint4 *>(row + c) = q;
                }
            }
        }
        if (have && gl == G - 1 && EMDEE_BOUND(BS_TYPED_ROW, p, a.n)) {
            const unsigned need = (unsigned)(S1 + total1);           // slots the row takes (total1 == 0: just the first segment)
            const bool fits = need <= ustride && total0 <= segcap && total1 <= segcap;
            a.cnt[p] = (act && fits) ? (total0 | (total1 << 16)) : 0;
            // (a segment longer than the row buffer asks for a longer stride like a row that does not fit: the buffer follows it)
            if (!fits) atomicMax(&a.flags[0], (int)max(max(need, ustride + 1u), 2u * (unsigned)max(total0, total1) + (unsigned)BLKL));
        }
    }
}

// ------------------------------------------------------------------------------------ force / stats / fused step
// k_brick's single-species coordinate-plane kernel run over the two segments of a row, each with its own pair constants.
template <typename real, class Shape, int THREADS, int G, int MODE, int BITMASK>
__global__ __launch_bounds__(THREADS, (Shape::NOC == 8 ? 4 : 1)) void k_typed(BrickArgs<real> a) {   // (2 x 2 x 2 bricks: <= 128 VGPRs, two workgroups per CU)
    constexpr int NGROUPS = (THREADS / WAVE) * (WAVE / G), BLK = EPL * G, NTC = Shape::NTC, NOC = Shape::NOC;
    constexpr int PITCH = typed_pitch<real, Shape, THREADS>(), PLANE_BYTES = PITCH * (int)sizeof(real);
    extern __shared__ __attribute__((aligned(16))) unsigned char s_dyn[];
    real *plane = reinterpret_cast<real *>(s_dyn);                          // x | y | z, PITCH records apart
    const unsigned char *plane_b = s_dyn;
    constexpr size_t TILE_BYTES = ((size_t)3 * PITCH * sizeof(real) + 15) & ~(size_t)15;
    TypedTables<Shape, THREADS> T;
    T.carve(s_dyn + TILE_BYTES);
    struct PairC { real sig2, e4; LJSeg<real> seg; };               // (seg: the force-only launches' folded constants, lj_pair.hpp)
    PairC *ptab = reinterpret_cast<PairC *>(s_dyn + TILE_BYTES + TypedTables<Shape, THREADS>::bytes(a.own_cap));   // [species_i * 2 + species_j]
    if (MODE == BRICK_STEP && a.guard != nullptr && *a.guard != 0) {
        if (threadIdx.x == 0) *a.trigger = 1;                               // a step queued behind a rebuild request: no trace
        return;
    }
    int bxi, byi, bzi, tile_n, n_own;
    if (!typed_setup<real, Shape, THREADS>(a, T, bxi, byi, bzi, tile_n, n_own)) return;
    const int tid = threadIdx.x, lane = tid & (WAVE - 1);

    int own_p[OWN_REGS], own_ti[OWN_REGS], own_m[OWN_REGS], own_q[OWN_REGS];
#pragma unroll
    for (int k = 0; k < OWN_REGS; k++) {
        const int o = tid + k * THREADS;
        own_p[k] = own_ti[k] = own_m[k] = own_q[k] = 0;
        if (o < n_own) {
            own_q[k] = typed_locate(T, o, own_ti[k], own_p[k]);
            own_m[k] = a.cnt[own_p[k]];
        }
    }
    // fp32 boxes: brick-relative tile coordinates (brick.hpp k_brick)
    constexpr bool REL = sizeof(real) == 4;
    __shared__ float s_relc[REL ? Shape::TX + Shape::TY + Shape::TZ : 1];   // cell-relative records: brick.hpp rel_cell_const
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
    typed_for_each_slot(T, [&](int s, int tc, int tx, int ty, int tz) {
        const int gp = T.gbeg[tc] + (s - T.off[tc]);
        const int sh = T.shift[tc % NTC];
        Rec<real> r = a.rec[gp];
        if (REL && a.rel) {                                   // cell-relative records: fixed-point tile coordinates (brick.hpp k_brick)
            r.x = (real)rel_tile(r.x, s_relc[tx]); r.y = (real)rel_tile(r.y, s_relc[Shape::TX + ty]);
            r.z = (real)rel_tile(r.z, s_relc[Shape::TX + Shape::TY + tz]);
        } else if (REL) {
            r.x = (real)(((double)r.x + (double)((sh & 3) - 1) * (double)a.g.len[0]) - org[0]);
            r.y = (real)(((double)r.y + (double)(((sh >> 2) & 3) - 1) * (double)a.g.len[1]) - org[1]);
            r.z = (real)(((double)r.z + (double)(((sh >> 4) & 3) - 1) * (double)a.g.len[2]) - org[2]);
        } else if (sh != (1 | (1 << 2) | (1 << 4))) {
            r.x += (real)((sh & 3) - 1) * a.g.len[0];
            r.y += (real)(((sh >> 2) & 3) - 1) * a.g.len[1];
            r.z += (real)(((sh >> 4) & 3) - 1) * a.g.len[2];
        }
        if (EMDEE_BOUND(BS_TYPED_TILE, s, PITCH)) { plane[s] = r.x; plane[PITCH + s] = r.y; plane[2 * PITCH + s] = r.z; }
    });
    if (tid == 0) {   // the sentinel record every unused row entry points at: fails r2 < rc2, never NaN
        const real big = sizeof(real) == 8 ? (real)1e30 : (real)1e18;
        plane[0] = big; plane[PITCH] = big; plane[2 * PITCH] = big;
    }
    if (tid < TNT * TNT) { PairC c; c.sig2 = a.tsig2[tid]; c.e4 = a.te4[tid]; c.seg = make_seg(a.model, c.sig2, c.e4); ptab[tid] = c; }
    // per own atom: {cell-order slot | species << 30, tile slot} and the two segment lengths n0 | n1 << 16
    T.carve_counts(a.own_cap);
#pragma unroll
    for (int k = 0; k < OWN_REGS; k++) {
        const int o = tid + k * THREADS;
        if (o < n_own && EMDEE_BOUND(BS_TYPED_OWN, o, a.own_cap)) { T.oinfo[o] = make_int2(own_p[k] | ((own_q[k] / NOC) << 30), own_ti[k]); T.ocnt[o] = own_m[k]; }
    }
    for (int o = tid + OWN_REGS * THREADS; o < n_own; o += THREADS) {
        int ti, p;
        const int q = typed_locate(T, o, ti, p);
        if (!EMDEE_BOUND(BS_TYPED_OWN, o, a.own_cap)) continue;
        T.oinfo[o] = make_int2(p | ((q / NOC) << 30), ti);
        T.ocnt[o] = a.cnt[p];
    }
    __syncthreads();

    LJModel<real> mdl = a.model;
    // fp64: 4 eps_ij folded into the segment's switch constants (lj_pair.hpp LJSeg: 3.077 -> 3.042 ms per launch, same box twice;
    // fp32 loses with it, 2.12 -> 2.21 ms: ten more live registers per lane; profiles/r05/call_v.sh)
#ifdef EMDEE_TYPED_NO_FOLD
    constexpr bool FOLD = false;
#else
    constexpr bool FOLD = sizeof(real) == 8;
#endif
    if (BITMASK == EMDEE_FORCES && FOLD) asm volatile("" : "+v"(mdl.nx0), "+v"(mdl.idl2));    // (the Horner constants are the segment's)
    else if (BITMASK == EMDEE_FORCES) asm volatile("" : "+v"(mdl.nx0), "+v"(mdl.idl2), "+v"(mdl.h4), "+v"(mdl.h3), "+v"(mdl.k6));
    else asm volatile("" : "+v"(mdl.x0), "+v"(mdl.k3));

    const int gl = lane & (G - 1);
    const int gid = (tid / WAVE) * (WAVE / G) + lane / G;
    unsigned long long st_entries = 0, st_inside = 0;
    int st_max = 0;
    // the first NPF blocks of a row (= of its species-0 segment) are prefetched one atom ahead
    constexpr int NPF = typed_prefetch_blocks(G, THREADS, Shape::NOC);
    struct IdxBuf { uint4 q[NPF]; };
    auto fetch = [&](int o) {
        IdxBuf b;
        const unsigned short *row = a.nbr + (size_t)(T.oinfo[min(o, n_own - 1)].x & 0x3fffffff) * a.stride + gl * EPL;
#pragma unroll
        for (int k = 0; k < NPF; k++) b.q[k] = *reinterpret_cast<const uint4 *>(row + k * BLK);   // stride >= (NPF + 1) BLK (host)
        return b;
    };
    IdxBuf nxt = fetch(gid);
    for (int ob = 0; ob < n_own; ob += NGROUPS) {             // wave-uniform trip count
        const int o = ob + gid;
        const bool have = o < n_own;
        const int2 info = have ? T.oinfo[o] : make_int2(0, 0);
        const int p = info.x & 0x3fffffff, t_i = (int)((unsigned)info.x >> 30), ti = info.y;
        const int mm = have ? T.ocnt[o] : 0, n0 = mm & 0xffff, n1 = (int)((unsigned)mm >> 16);
        const int S1 = (n0 + BLK - 1) / BLK * BLK;
        const IdxBuf cur = nxt;
        const unsigned short *rowp = a.nbr + (size_t)p * a.stride + gl * EPL;
        // the species-1 segment's first block is requested now and arrives while segment 0 is being worked on
        uint4 seg1 = make_uint4(0, 0, 0, 0);
        if (S1 < a.stride) seg1 = *reinterpret_cast<const uint4 *>(rowp + S1);
        nxt = fetch(o + NGROUPS);
        const int wm0 = wave_group_max<G>(n0), wm1 = wave_group_max<G>(n1);
        const real xi = plane[ti], yi = plane[PITCH + ti], zi = plane[2 * PITCH + ti];
        const PairC c0 = ptab[t_i * TNT], c1 = ptab[t_i * TNT + 1];
        real fx = 0, fy = 0, fz = 0, e = 0, w = 0;
        // one block of 8 G neighbours of one species: lane gl holds entries b0 + gl + t G, t = 0..7, in q
        auto block = [&](const uint4 &q, int b0, int wm, int m, const PairC &c) {
            // The coordinates of entry t + 1 are requested BEFORE the arithmetic of entry t (round 5): this kernel runs four
            // wavefronts per SIMD (one 1024-thread workgroup per CU: the rc = 3.5 sigma tile fills the LDS), too few to hide
            // the three dependent LDS reads of every pair step behind other waves -- SQ counters: the vector units were busy
            // 70 % of the kernel against 87 % in the single-species kernel at six waves (profiles/r05/valu_f64_mix_rc3.5.txt).
            // Every entry of a block is readable whatever the trip count (sentinels pad a segment): the read ahead is safe.
            const unsigned char *pn = plane_b + pick16(q, 0);
            real xn = *reinterpret_cast<const real *>(pn), yn = *reinterpret_cast<const real *>(pn + PLANE_BYTES),
                 zn = *reinterpret_cast<const real *>(pn + 2 * PLANE_BYTES);
#pragma unroll
            for (int t = 0; t < EPL; t++) {
                if (b0 + t * G >= wm) break;                  // wave-uniform; entries past a segment's end are sentinels
                const real xj = xn, yj = yn, zj = zn;
                if (t + 1 < EPL) {
                    const unsigned char *pj = plane_b + pick16(q, t + 1);    // byte offset: three reads off one address register
                    xn = *reinterpret_cast<const real *>(pj);
                    yn = *reinterpret_cast<const real *>(pj + PLANE_BYTES);
                    zn = *reinterpret_cast<const real *>(pj + 2 * PLANE_BYTES);
                }
                const real dx = xi - xj, dy = yi - yj, dz = zi - zj;
                const real r2 = dx * dx + dy * dy + dz * dz;
                if (MODE == BRICK_STATS) {
                    st_inside += (b0 + t * G + gl < m && r2 < a.model.rc2) ? 1ull : 0ull;
                } else if (r2 < a.model.rc2) {                // strict test (Q2)
                    const real inv_r2 = fast_rcp(r2);
                    if (BITMASK == EMDEE_FORCES) {
                        // (A/B: profiles/build_variant.sh nofold -DEMDEE_TYPED_NO_FOLD=1)
                        const real wr2 = FOLD ? lj_force_over_r2_seg(r2, inv_r2, mdl, c.seg) : lj_force_over_r2(r2, inv_r2, mdl, c.sig2, c.e4);
                        fx += wr2 * dx; fy += wr2 * dy; fz += wr2 * dz;
                    } else {
                        real E, W;
                        lj_interaction_pair(r2, inv_r2, mdl, c.sig2, c.e4, E, W);
                        if (BITMASK & EMDEE_FORCES) {
                            const real wr2 = W * inv_r2;          // src/nonbonded.jl:139
                            fx += wr2 * dx; fy += wr2 * dy; fz += wr2 * dz;
                        }
                        if (BITMASK & EMDEE_ENERGIES) e += E;
                        if (BITMASK & EMDEE_VIRIALS) w += W;
                    }
                }
            }
        };
        // ---- segment 0: neighbours of species 0.  The trip bound is the wavefront's longest segment; a lane whose own
        // segment ended before block k must not walk into its species-1 entries, which start there: it gets sentinels
        // (one select per block of 8 pair steps)
        const uint4 none = make_uint4(0, 0, 0, 0);
#pragma unroll
        for (int k = 0; k < NPF; k++)
            if (k * BLK < wm0) block(k * BLK < S1 ? cur.q[k] : none, k * BLK, wm0, n0, c0);
        for (int b0 = NPF * BLK; b0 < wm0; b0 += BLK) {       // (more than NPF blocks of one species: on demand)
            uint4 q = none;
            if (b0 < S1) q = *reinterpret_cast<const uint4 *>(rowp + b0);   // (S1 <= stride)
            block(q, b0, wm0, n0, c0);
        }
        // ---- segment 1: neighbours of species 1, one block ahead
        {
            uint4 q = seg1;
            for (int b0 = 0; b0 < wm1; b0 += BLK) {
                uint4 more = make_uint4(0, 0, 0, 0);
                if (b0 + BLK < wm1 && S1 + b0 + BLK < a.stride) more = *reinterpret_cast<const uint4 *>(rowp + S1 + b0 + BLK);
                block(q, b0, wm1, n1, c1);
                q = more;
            }
        }
        if (MODE == BRICK_STATS) {
            if (gl == 0) { st_entries += (unsigned long long)(n0 + n1); st_max = max(st_max, n0 + n1); }
        } else {
            if (BITMASK & EMDEE_FORCES) { fx = group_sum_to_last<G>(fx); fy = group_sum_to_last<G>(fy); fz = group_sum_to_last<G>(fz); }
            if (BITMASK & EMDEE_ENERGIES) e = group_sum_to_last<G>(e);
            if (BITMASK & EMDEE_VIRIALS) w = group_sum_to_last<G>(w);
            if (MODE == BRICK_STEP) {
                if (have && gl == G - 1) {
                    real vx = a.vel[p], vy = a.vel[a.pitch + p], vz = a.vel[2 * a.pitch + p];
                    const real bx = a.xb[p], by = a.xb[a.pitch + p], bz = a.xb[2 * a.pitch + p];
                    real imv = 1, nx = 0, ny = 0, nz = 0;
                    if (a.inv_mass) imv = a.inv_mass[p];
                    if (a.noise) { nx = a.noise[p]; ny = a.noise[a.pitch + p]; nz = a.noise[2 * a.pitch + p]; }
                    const real cm = a.kick_c * imv;
                    vx += cm * fx; vy += cm * fy; vz += cm * fz;
                    vx = a.lgv_c1 * vx + nx; vy = a.lgv_c1 * vy + ny; vz = a.lgv_c1 * vz + nz;   // (NVE: c1 = 1, noise 0: bit for bit)
                    a.vel_next[p] = vx; a.vel_next[a.pitch + p] = vy; a.vel_next[2 * a.pitch + p] = vz;
                    Rec<real> r = a.rec[p];                    // keeps the LJAtom fields bit for bit
                    r.x += a.dt * vx; r.y += a.dt * vy; r.z += a.dt * vz;
                    a.rec_next[p] = r;
                    const real ex = r.x - bx, ey = r.y - by, ez = r.z - bz;
                    if (ex * ex + ey * ey + ez * ez > a.thr2) *a.trigger = 1;
                }
            } else if (have && gl == G - 1) {
                if (a.user_f != nullptr || a.user_e != nullptr || a.user_w != nullptr) {
                    const size_t i = (size_t)a.perm[p];                     // caller index of this atom
                    if ((BITMASK & EMDEE_FORCES) && a.user_f) { a.user_f[3 * i] = fx; a.user_f[3 * i + 1] = fy; a.user_f[3 * i + 2] = fz; }
                    if ((BITMASK & EMDEE_ENERGIES) && a.user_e) a.user_e[i] = (real)0.5 * e;
                    if ((BITMASK & EMDEE_VIRIALS) && a.user_w) a.user_w[i] = (real)0.5 * w;
                } else {
                    if (BITMASK & EMDEE_FORCES) { a.frc[p] = fx; a.frc[a.pitch + p] = fy; a.frc[2 * a.pitch + p] = fz; }
                    if (BITMASK & EMDEE_ENERGIES) a.en[p] = (real)0.5 * e;   // src/nonbonded.jl:142-145
                    if (BITMASK & EMDEE_VIRIALS) a.vir[p] = (real)0.5 * w;
                }
            }
        }
    }
    if (MODE == BRICK_STATS) {
#pragma unroll
        for (int off = 1; off < WAVE; off <<= 1) {
            st_entries += __shfl_xor(st_entries, off);
            st_inside += __shfl_xor(st_inside, off);
            st_max = max(st_max, __shfl_xor(st_max, off));
        }
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

// verification accessor: the neighbour rows as CALLER ids, species-0 neighbours first
template <typename real, class Shape, int THREADS, int G>
__global__ __launch_bounds__(THREADS) void k_typed_export(BrickArgs<real> a, int *__restrict__ counts, int *__restrict__ out, int capacity,
                                                          const int *__restrict__ cmap = nullptr) {
    constexpr int BLK = EPL * G, NTT = TNT * Shape::NTC;
    extern __shared__ __attribute__((aligned(16))) unsigned char s_dyn[];
    TypedTables<Shape, THREADS> T;
    T.carve(s_dyn);
    int bxi, byi, bzi, tile_n, n_own;
    if (!typed_setup<real, Shape, THREADS>(a, T, bxi, byi, bzi, tile_n, n_own)) return;
    for (int o = threadIdx.x; o < n_own; o += THREADS) {
        int ti, p;
        typed_locate(T, o, ti, p);
        if (a.perm[p] >= a.n_owned) continue;
        const int i = cmap ? cmap[a.perm[p]] : a.perm[p];
        const int m = a.cnt[p], n0 = m & 0xffff, n1 = (int)((unsigned)m >> 16), S1 = (n0 + BLK - 1) / BLK * BLK;
        counts[i] = n0 + n1;
        const unsigned short *row = a.nbr + (size_t)p * a.stride;
        for (int e = 0; e < n0 + n1 && e < capacity; e++) {
            const unsigned pos = e < n0 ? row_position<G>((unsigned)e) : (unsigned)S1 + row_position<G>((unsigned)(e - n0));
            const int sl = (int)row[pos] >> a.idx_shift;
            int lo = 0, hi = NTT;                             // typed tile cell with off[tc] <= sl < off[tc + 1]
            while (hi - lo > 1) {
                const int mid = (lo + hi) >> 1;
                if (T.off[mid] <= sl) lo = mid; else hi = mid;
            }
            const int j = a.perm[T.gbeg[lo] + (sl - T.off[lo])];
            out[(size_t)i * capacity + e] = cmap ? cmap[j] : j;
        }
    }
}

}  // namespace emdee
