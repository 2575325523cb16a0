// brick_tbuild.hpp -- the TRANSPOSED neighbour build (round 4): candidates in the lanes, own atoms in the loop.
//
// k_brick_build (brick.hpp) gives every own atom a group of 8 lanes; each lane walks a share of the atom's candidates
// (one ds_read_b128 + 8.5 VALU instructions per candidate), records one bit per candidate, and the hits are then
// extracted from the bit strings by the lane that found them -- emission loops that run as long as the busiest of 64
// lanes (profiles/r03: 250 wave-instructions per 8 atoms where 85 would do), and ~400 instructions of per-row
// bookkeeping that every one of the 8 lanes repeats.  119 VALU wave-instructions per atom, all told, on a kernel that
// sits on its VALU issue floor.
//
// Here the roles are swapped.  A wavefront takes one OWN CELL at a time.  The candidates of all its atoms are the same
// nine runs of tile slots (3 cells along x for each (dy, dz)); they are concatenated (~475 slots at rho* = 0.8,
// r_list = 2.8) and dealt to the lanes in SEGMENTS of 64: lane l holds candidate 64 j + l of segment j IN REGISTERS, two
// segments per register pair.  Then the cell's atoms go by one at a time: the atom's position is one broadcast LDS
// read, the distance test of 128 candidates is six packed fp32 instructions (v_pk_add_f32 / v_pk_fma_f32: the only fp32
// VALU form that does two operations per lane on gfx950) and two compares whose results ARE the hit masks, as SCALARS.
// The hits are then stored by the lanes that hold them, at base + (set bits below my lane): v_mbcnt_lo / v_mbcnt_hi /
// v_add_lshl_u32 and one masked 2-byte LDS store per segment -- no bit strings, no extraction loop, nothing that
// depends on the busiest lane, and the row bookkeeping is scalar (SALU).  Rows are assembled in LDS in plain order,
// a few atoms at a time, and flushed in the lane-major block layout the force kernels read (row_position<GL>), as
// coalesced 16-byte stores.
//
// What it costs: no x sub-bins (a lane cannot skip a candidate the other atoms of the cell need: 475 candidates per
// atom instead of 356), 7 % of the lane slots idle in the last segment.  What it saves: everything else.
// Same neighbour SET as the other builds, by the same argument: fp32 test of brick-relative coordinates, exact fp64
// re-test of the pairs inside the proven rounding band (fp64 boxes).
//
// Replaces: find_action_partners1! (src/cells.jl:224-297).
#pragma once

#include "brick.hpp"

namespace emdee {

constexpr int TB_SEG = WAVE;   // candidates per segment: one per lane

// rows assembled in LDS before they are flushed: as many as one pass of the wavefront can flush (one 16-byte chunk per lane)
__host__ __device__ inline int tb_rows(int stride) {
    const int chunks = stride / EPL;
    const int b = chunks >= WAVE ? 1 : WAVE / chunks;
    return b > 8 ? 8 : b;
}
// per-wavefront scratch: the candidate table of a cell (q -> tile slot, 2 bytes each), then that cell's row buffers
// (+ 128 bytes in front: one 2-byte dump slot per lane, where the lanes that hold no hit of a segment store)
constexpr int TB_DUMP = WAVE * 2;
template <int NPAIR>
__host__ __device__ inline size_t tb_wave_bytes(int stride) {
    const size_t cand = (size_t)NPAIR * 2 * TB_SEG * 2, rows = (size_t)tb_rows(stride) * (size_t)stride * 2;
    return (((cand > rows ? cand : rows) + 15) & ~(size_t)15) + TB_DUMP;
}
template <class Shape, int THREADS, int NPAIR>
static inline size_t brick_tbuild_lds_bytes(int tile_cap, int own_cap, int stride) {
    return (size_t)tile_cap * 16 + BrickTables<Shape, THREADS>::bytes(own_cap) + (size_t)(THREADS / WAVE) * tb_wave_bytes<NPAIR>(stride);
}

// GL = lanes per atom of the force kernels (fixes the row layout); NPAIR = segment pairs held in registers
// (a cell whose nine runs hold more than 128 NPAIR candidates raises flags[3]: the host falls back to k_brick_build)
#ifndef EMDEE_TB_WAVES
#define EMDEE_TB_WAVES 6
#endif
template <typename real, class Shape, int THREADS, int GL, int NPAIR>
__global__ __launch_bounds__(THREADS, (THREADS <= 512 ? EMDEE_TB_WAVES : 4)) void k_brick_build_t(BrickArgs<real> a) {
    constexpr int BX = Shape::BX, BY = Shape::BY, TX = Shape::TX, TY = Shape::TY, NTC = Shape::NTC, NOC = Shape::NOC;
    constexpr int NWAVES = THREADS / WAVE, NSEG = 2 * NPAIR;
    constexpr bool BAND = sizeof(real) == 8;              // fp32 boxes: the fp32 test is the definition of the set
    extern __shared__ __attribute__((aligned(16))) unsigned char s_dyn[];
    float4 *tile = reinterpret_cast<float4 *>(s_dyn);     // {x, y, z relative to the brick origin, cell-order slot}
    BrickTables<Shape, THREADS> T;
    T.carve(s_dyn + (size_t)a.tile_cap * 16);
    int bxi, byi, bzi, tile_n, n_own;
    if (!brick_setup<real, Shape, THREADS>(a, T, bxi, byi, bzi, tile_n, n_own)) return;
    auto sc = [](int v) { return __builtin_amdgcn_readfirstlane(v); };   // a value every lane holds, as a scalar
    const int tid = threadIdx.x, lane = tid & (WAVE - 1), wv = sc(tid / WAVE);

    // brick origin: fp64 boxes are re-based here so that fp32 coordinates stay small
    real org[3] = {0, 0, 0};
    if (sizeof(real) == 8) {
        org[0] = a.g.lo[0] + (real)(bxi * BX) * (a.g.len[0] / (real)a.g.M[0]);
        org[1] = a.g.lo[1] + (real)(byi * BY) * (a.g.len[1] / (real)a.g.M[1]);
        org[2] = a.g.lo[2] + (real)(bzi * Shape::BZ) * (a.g.len[2] / (real)a.g.M[2]);
    }
    // decomposed runs: ghosts own no row (one byte per own atom, in the table's own-atom area)
    const bool all_owned = !a.any_ghosts;
    unsigned char *ownflag = reinterpret_cast<unsigned char *>(T.oinfo);
    if (!all_owned) {
        for (int o = tid; o < n_own; o += THREADS) {
            int ti, p;
            brick_locate(T, o, ti, p);
            ownflag[o] = a.perm[p] < a.n_owned ? 1 : 0;
        }
    }
    brick_for_each_slot(T, [&](int s, int tc, int, int, int) {
        const int gp = T.gbeg[tc] + (s - T.off[tc]);
        const int sh = T.shift[tc];
        const Rec<real> r = a.rec[gp];
        float4 q;
        q.x = (float)((r.x + (real)((sh & 3) - 1) * a.g.len[0]) - org[0]);
        q.y = (float)((r.y + (real)(((sh >> 2) & 3) - 1) * a.g.len[1]) - org[1]);
        q.z = (float)((r.z + (real)(((sh >> 4) & 3) - 1) * a.g.len[2]) - org[2]);
        q.w = __int_as_float(gp);
        if (EMDEE_BOUND(BS_BUILD_TILE, s, a.tile_cap)) tile[s] = q;
    });
    __syncthreads();

    // ---- from here on the wavefronts are on their own: no barrier, every LDS exchange stays inside a wavefront ----
    unsigned short *const lds16 = reinterpret_cast<unsigned short *>(s_dyn);
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char *)s_dyn;   // LDS byte address of s_dyn
    const unsigned wscr_i0 = (unsigned)(((size_t)a.tile_cap * 16 + BrickTables<Shape, THREADS>::bytes(a.own_cap) + (size_t)wv * tb_wave_bytes<NPAIR>(a.stride)) / 2);
    const unsigned wscr_i = wscr_i0 + TB_DUMP / 2;
    const unsigned dump_addr = lds0 + 2u * (wscr_i0 + (unsigned)lane);   // byte address of this lane's dump slot
    unsigned short *const wscr = lds16 + wscr_i;           // this wavefront's scratch
    const int stride = a.stride, chunks = stride / EPL, B = tb_rows(stride);
    // the 16-byte chunk of a batch this lane flushes: batch row and chunk of that row (B > 1: B chunks <= 64, one trip)
    const int f_rb = B > 1 ? lane / chunks : 0, f_cc = B > 1 ? lane - f_rb * chunks : lane;
    constexpr unsigned BLKL = EPL * GL;                    // entries per lane-major block of the force kernels' rows
    // ... reads the entries f_src + GL t (t = 0..7) of the row buffers and writes 8 entries at f_dst of the rows (B > 1; one trip)
    const unsigned f_c = (unsigned)f_cc * EPL;
    const unsigned f_src = (unsigned)(f_rb * stride) + (f_c / BLKL) * BLKL + (f_c % BLKL) / EPL, f_dst = (unsigned)(f_rb * stride) + f_c;
    float nrl2 = -(float)a.rlist2, margin_v = a.margin;
    asm volatile("" : "+v"(nrl2), "+v"(margin_v));
    const float BIG = 1.0e18f;                             // coordinates of a lane that holds no candidate

    for (int oc = wv; oc < NOC; oc += NWAVES) {
        const int ox = oc % BX, oy = (oc / BX) % BY, oz = oc / (BX * BY);
        const int tcm = (ox + 1) + TX * ((oy + 1) + TY * (oz + 1));
        const int o0 = sc(T.own[oc]), n_cell = sc(T.own[oc + 1]) - o0;   // 0: not a cell of this brick, or nothing but ghosts
        if (n_cell <= 0) continue;
        const int ti0 = sc(T.off[tcm]), p0 = sc(T.gbeg[tcm]);
        // The candidate runs, as runs of tile slots, in the order they are dealt to the lanes: the own cell FIRST (so that atom i of
        // the cell is candidate i: its own bit sits in segment i / 64, no search), then the two other cells of its tile row, then
        // the eight other tile rows (dy, dz), three cells along x each.  Lane r < NRUN holds run r; a prefix gives its first q.
        constexpr int NRUN = 11;
        int c0 = 0, span = 0;
        if (lane < NRUN) {
            if (lane < 3) {
                const int tc = lane == 0 ? tcm : (lane == 1 ? tcm - 1 : tcm + 1);
                c0 = T.off[tc];
                span = T.off[tc + 1] - c0;
            } else {
                const int r = lane - 3 + (lane >= 7 ? 1 : 0);             // 0..8 without 4
                const int tcr = ox + TX * ((oy + r % 3) + TY * (oz + r / 3));
                c0 = T.off[tcr];
                span = T.off[tcr + 3] - c0;
            }
        }
        int incl = span;
        incl += __builtin_amdgcn_update_dpp(0, incl, DPP_ROW_SHR1, 0xf, 0xf, true);
        incl += __builtin_amdgcn_update_dpp(0, incl, DPP_ROW_SHR2, 0xf, 0xf, true);
        incl += __builtin_amdgcn_update_dpp(0, incl, DPP_ROW_SHR4, 0xf, 0xf, true);
        incl += __builtin_amdgcn_update_dpp(0, incl, DPP_ROW_SHR8, 0xf, 0xf, true);
        const int pex = incl - span;
        const int Q = __builtin_amdgcn_readlane(incl, NRUN - 1);
        const int nseg = (Q + TB_SEG - 1) / TB_SEG;
        if (nseg > NSEG || n_cell > 2 * TB_SEG) {           // the host builds again with k_brick_build
            if (lane == 0) atomicMax(&a.flags[3], Q);
            continue;
        }
        // candidate table: q -> tile slot (the runs one after the other)
#pragma unroll 1
        for (int r = 0; r < NRUN; r++) {
            const int c0r = __builtin_amdgcn_readlane(c0, r), spr = __builtin_amdgcn_readlane(span, r), pr = __builtin_amdgcn_readlane(pex, r);
            for (int k = lane; k < spr; k += WAVE)
                if (EMDEE_BOUND(BS_TBUILD_CAND, pr + k, NSEG * TB_SEG)) wscr[pr + k] = (unsigned short)(c0r + k);
        }
        // my candidates: segment j, lane l <-> candidate 64 j + l
        f32x2 X[NPAIR], Y[NPAIR], Z[NPAIR];
        unsigned svp[NPAIR];                                 // their list entries (tile slot << idx_shift), two per register
#pragma unroll
        for (int j = 0; j < NSEG; j++) {
            float x = BIG, y = BIG, z = BIG;
            unsigned s = 0;
            if (j < nseg) {
                const int q = j * TB_SEG + lane;
                if (q < Q) {
                    s = wscr[q];
                    const float4 c = tile[s];
                    x = c.x; y = c.y; z = c.z;
                }
            }
            X[j / 2][j & 1] = x; Y[j / 2][j & 1] = y; Z[j / 2][j & 1] = z;
            if (j & 1) svp[j / 2] |= (s << a.idx_shift) << 16;
            else svp[j / 2] = s << a.idx_shift;
        }
        // the scratch now becomes the row buffers: SENTINEL slot 0 everywhere (the force kernels walk whole blocks)
        for (int c = lane * EPL; c < B * stride; c += WAVE * EPL) *reinterpret_cast<uint4 *>(wscr + c) = make_uint4(0, 0, 0, 0);

        int nb = 0, lens = 0, pfirst = p0;
        auto flush = [&]() {
            auto chunk = [&](const unsigned short *src, unsigned short *dst) {
                uint4 q;
                q.x = (unsigned)src[0 * GL] | ((unsigned)src[1 * GL] << 16);
                q.y = (unsigned)src[2 * GL] | ((unsigned)src[3 * GL] << 16);
                q.z = (unsigned)src[4 * GL] | ((unsigned)src[5 * GL] << 16);
                q.w = (unsigned)src[6 * GL] | ((unsigned)src[7 * GL] << 16);
                *reinterpret_cast<uint4 *>(dst) = q;
            };
            unsigned short *rows = a.nbr + (size_t)pfirst * stride;
            if (!EMDEE_BOUND(BS_BUILD_ROW, pfirst + nb - 1, a.n)) return;
            if (B > 1) {
                if (f_rb < nb) chunk(wscr + f_src, rows + f_dst);
            } else {
                for (unsigned c = (unsigned)lane * EPL; c < (unsigned)stride; c += WAVE * EPL)
                    chunk(wscr + (c / BLKL) * BLKL + (c % BLKL) / EPL, rows + c);
            }
            for (int c = lane * EPL; c < nb * stride; c += WAVE * EPL) *reinterpret_cast<uint4 *>(wscr + c) = make_uint4(0, 0, 0, 0);
            if (lane < nb) a.cnt[pfirst + lane] = lens;
        };

        // The atoms of the cell, with NP register pairs in use (the instantiation with the fewest pairs that hold the cell's
        // candidates: no per-pair test of the segment count inside the loop; a pair past the last segment holds far-away
        // coordinates and its masks come out empty).
        auto run_atoms = [&](auto NPt) {
            constexpr int NP = decltype(NPt)::value, NS = 2 * NP;
#pragma unroll 1
            for (int i = 0; i < n_cell; i++) {
                const int p = p0 + i;
                const float4 qi = tile[ti0 + i];                 // one address for all lanes: a broadcast read
                const f32x2 qx = {qi.x, qi.x}, qy = {qi.y, qi.y}, qz = {qi.z, qi.z};
                bool owned = true;
                if (!all_owned) owned = sc((int)ownflag[o0 + i]) != 0;    // ghosts own no row (an empty one is written)
                unsigned long long m[NS];
                // t = d^2 - r_list^2 of my candidates, two segments per packed instruction; the masks of the listed ones.
                // FIX 1: which of my candidates are inside the rounding band of the fp32 test (bit j of `band`); FIX 2: those
                // take the exact answer (bit j of `pass`).
                auto masks = [&](auto FIX, unsigned &band, unsigned pass) -> float {
                    float tmin = 3.0e38f;
#pragma unroll
                    for (int jp = 0; jp < NP; jp++) {
                        const f32x2 dx = qx - X[jp], dy = qy - Y[jp], dz = qz - Z[jp];
                        f32x2 t = __builtin_elementwise_fma(dx, dx, f32x2{nrl2, nrl2});
                        t = __builtin_elementwise_fma(dy, dy, t);
                        t = __builtin_elementwise_fma(dz, dz, t);
                        if constexpr (decltype(FIX)::value == 1) {
                            band |= (__builtin_fabsf(t.x) <= margin_v ? 1u : 0u) << (2 * jp);
                            band |= (__builtin_fabsf(t.y) <= margin_v ? 1u : 0u) << (2 * jp + 1);
                        } else if constexpr (decltype(FIX)::value == 2) {
                            if ((band >> (2 * jp)) & 1u) t.x = ((pass >> (2 * jp)) & 1u) ? -1.f : 1.f;
                            if ((band >> (2 * jp + 1)) & 1u) t.y = ((pass >> (2 * jp + 1)) & 1u) ? -1.f : 1.f;
                        } else if constexpr (BAND) {
                            asm("v_min3_f32 %0, %0, |%1|, |%2|" : "+v"(tmin) : "v"(t.x), "v"(t.y));
                        }
                        if constexpr (decltype(FIX)::value != 1) {
                            m[2 * jp] = __builtin_amdgcn_ballot_w64(t.x < 0.f);
                            m[2 * jp + 1] = __builtin_amdgcn_ballot_w64(t.y < 0.f);
                        }
                    }
                    return tmin;
                };
                unsigned band = 0;
#ifdef EMDEE_TB_ABLATE
                float tmin = 1.0f;
                if (EMDEE_TB_ABLATE & 2) { for (int j = 0; j < NS; j++) m[j] = 0x0101010101010101ull << (j & 7); }
                else tmin = masks(std::integral_constant<int, 0>{}, band, 0u);
#else
                const float tmin = masks(std::integral_constant<int, 0>{}, band, 0u);
#endif
                if constexpr (BAND) {
                    if (__builtin_expect(__builtin_amdgcn_ballot_w64(tmin <= margin_v) != 0, 0)) {
                        // rare (3 atoms in 1000): the pairs inside the rounding band are decided with the exact fp64 records, one per
                        // lane and trip -- ONE copy of that code, not one per candidate register
                        masks(std::integral_constant<int, 1>{}, band, 0u);
                        unsigned pass = 0, todo = band;
                        while (__builtin_amdgcn_ballot_w64(todo != 0u) != 0) {
                            if (todo != 0u) {
                                const int j = __ffs((int)todo) - 1;
                                todo &= todo - 1u;
                                const int q = j * TB_SEG + lane;          // its tile slot, from the run table (lanes 0..NRUN-1)
                                int adj = 0;
#pragma unroll 1
                                for (int r = 0; r < NRUN; r++) {
                                    const int pr = __builtin_amdgcn_readlane(pex, r), cr = __builtin_amdgcn_readlane(c0, r);
                                    adj = q >= pr ? cr - pr : adj;
                                }
                                const int s = q + adj;
                                int lo = 0, hi = NTC;                    // tile cell with off[tc] <= s < off[tc + 1]
                                while (hi - lo > 1) {
                                    const int mid = (lo + hi) >> 1;
                                    if (T.off[mid] <= s) lo = mid; else hi = mid;
                                }
                                const int sh = T.shift[lo];
                                const Rec<real> ri = a.rec[p], rj = a.rec[T.gbeg[lo] + (s - T.off[lo])];
                                const real ex = ri.x - (rj.x + (real)((sh & 3) - 1) * a.g.len[0]);
                                const real ey = ri.y - (rj.y + (real)(((sh >> 2) & 3) - 1) * a.g.len[1]);
                                const real ez = ri.z - (rj.z + (real)(((sh >> 4) & 3) - 1) * a.g.len[2]);
                                pass |= (ex * ex + ey * ey + ez * ez < a.rlist2 ? 1u : 0u) << j;
                            }
                        }
                        masks(std::integral_constant<int, 2>{}, band, pass);
                    }
                }
                // not the atom itself: candidate i
                {
                    const unsigned long long self = 1ull << (i & (TB_SEG - 1));
                    if (i < TB_SEG) m[0] &= ~self; else m[1] &= ~self;
                }
                int total = 0;
#pragma unroll
                for (int j = 0; j < NS; j++) total += __builtin_popcountll(m[j]);
                if (!owned) total = 0;
                const bool over = total > stride;                 // the host sees the count, grows the stride and builds again
                if (__builtin_expect(over, 0)) {
                    if (lane == 0) atomicMax(&a.flags[0], total);
                    total = 0;
                }
#ifdef EMDEE_TB_ABLATE
                if (EMDEE_TB_ABLATE & 1) total = 0;
#endif
                if (total != 0) {
                    // Hits go to the row buffer in plain order: entry (hits of earlier segments) + (hits in lanes below mine).
                    // Every lane stores, the ones without a hit into their dump slot: no execution mask to set and restore per
                    // segment (each a scalar instruction between dependent vector ones: this loop is bound by the issue of
                    // dependent instructions, not by their number).  The scalar base is the LDS address in half-words, so that
                    // ONE v_add_lshl_u32 forms the byte address.
                    unsigned base = lds0 / 2u + wscr_i + (unsigned)(nb * stride);
#pragma unroll
                    for (int j = 0; j < NS; j++) {
                        const unsigned long long mm = m[j];
                        unsigned addr;
                        // (written out: left to itself the compiler narrows the execution mask around the address arithmetic
                        // again and keeps the high halves of the packed entries in registers of their own)
                        if (j & 1)
                            asm volatile("v_mbcnt_lo_u32_b32 %0, %2, 0\n\tv_mbcnt_hi_u32_b32 %0, %3, %0\n\tv_add_lshl_u32 %0, %0, %4, 1\n\t"
                                         "v_cndmask_b32 %0, %5, %0, %1\n\tds_write_b16_d16_hi %0, %6"
                                         : "=&v"(addr) : "s"(mm), "s"((unsigned)mm), "s"((unsigned)(mm >> 32)), "s"(base), "v"(dump_addr), "v"(svp[j / 2]) : "memory");
                        else
                            asm volatile("v_mbcnt_lo_u32_b32 %0, %2, 0\n\tv_mbcnt_hi_u32_b32 %0, %3, %0\n\tv_add_lshl_u32 %0, %0, %4, 1\n\t"
                                         "v_cndmask_b32 %0, %5, %0, %1\n\tds_write_b16 %0, %6"
                                         : "=&v"(addr) : "s"(mm), "s"((unsigned)mm), "s"((unsigned)(mm >> 32)), "s"(base), "v"(dump_addr), "v"(svp[j / 2]) : "memory");
                        base += (unsigned)__builtin_popcountll(mm);
                    }
                }
                lens = lane == nb ? total : lens;   // (row length of batch row nb, kept by lane nb)
                nb++;
                if (nb == B || i == n_cell - 1) {
#ifdef EMDEE_TB_ABLATE
                    if (!(EMDEE_TB_ABLATE & 4))
#endif
                    flush();
                    nb = 0;
                    pfirst = p + 1;
                }
            }
        };
        if (nseg <= 4 && NPAIR > 2) run_atoms(std::integral_constant<int, 2>{});
        else if (nseg <= 8 && NPAIR > 4) run_atoms(std::integral_constant<int, 4>{});
        else run_atoms(std::integral_constant<int, NPAIR>{});
    }
}

}  // namespace emdee
