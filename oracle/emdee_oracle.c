/*
 * emdee_oracle.c -- CPU restatement of EmDee.jl's nonbonded pair-force path.
 * TEST INFRASTRUCTURE ONLY; PARITY UNPINNED in the golden-vector sense -- see
 * emdee_oracle.h for what pins it instead.
 *
 * Build: make -C oracle   (gcc -O2 -ffp-contract=off -fopenmp -shared)
 */
#include "emdee_oracle.h"

#include <math.h>
#include <omp.h>
#include <stdlib.h>
#include <string.h>

int32_t orc_cells_per_dimension(double L, double cutoff, int32_t ndiv)
{
    return (int32_t)floor((double)ndiv * L / cutoff);          /* src/cells.jl:36 */
}

/* ---- float instantiation (the reference's own precision) ---- */
#define REAL float
#define SFX(n) n##_f32
#define MODEL orc_model32
#define RINT(x) rintf(x)
#define FLOOR(x) floorf(x)
#define SQRT(x) sqrtf(x)
#include "oracle_impl.inc"
#undef REAL
#undef SFX
#undef MODEL
#undef RINT
#undef FLOOR
#undef SQRT

/* ---- double instantiation (north-star fp64 path) ---- */
#define REAL double
#define SFX(n) n##_f64
#define MODEL orc_model64
#define RINT(x) rint(x)
#define FLOOR(x) floor(x)
#define SQRT(x) sqrt(x)
#include "oracle_impl.inc"
#undef REAL
#undef SFX
#undef MODEL
#undef RINT
#undef FLOOR
#undef SQRT

/* src/lennard_jones.jl:6-11: arithmetic in the argument type, fields stored Float32 */
void orc_model_f32(double cutoff, double sw, orc_model32 *m)
{
    m->rc2 = (float)(cutoff * cutoff);
    m->rs2 = (float)(sw * sw);
    m->inv_delta2 = (float)(1.0 / (cutoff * cutoff - sw * sw));
}

void orc_model_f64(double cutoff, double sw, orc_model64 *m)
{
    m->rc2 = cutoff * cutoff;
    m->rs2 = sw * sw;
    m->inv_delta2 = 1.0 / (cutoff * cutoff - sw * sw);
}

/* src/lennard_jones.jl:13 -- LJAtom(0.5 sigma, 2 sqrt(eps)), stored Float32 */
void orc_lj_atom(double eps, double sigma, orc_atom *a)
{
    a->half_sigma = (float)(0.5 * sigma);
    a->twice_sqrt_eps = (float)(2.0 * sqrt(eps));
}

void orc_interaction_f32(float r2, const orc_model32 *m, orc_atom ai, orc_atom aj, int mode, float *E, float *W)
{ interaction_f32(r2, m, ai, aj, mode, E, W); }

void orc_interaction_f64(double r2, const orc_model64 *m, orc_atom ai, orc_atom aj, int mode, double *E, double *W)
{ interaction_f64(r2, m, ai, aj, mode, E, W); }

void orc_naive_f32(int32_t N, const float *pos, float L, const orc_model32 *m, const orc_atom *atoms, int mode,
                   float *forces, float *energies, float *virials)
{ naive_f32(N, pos, L, m, atoms, mode, forces, energies, virials); }

void orc_naive_f64(int32_t N, const double *pos, double L, const orc_model64 *m, const orc_atom *atoms, int mode,
                   double *forces, double *energies, double *virials)
{ naive_f64(N, pos, L, m, atoms, mode, forces, energies, virials); }

int32_t orc_cells_f64(int32_t N, const double *pos, double L, double cutoff, int32_t ndiv,
                      int32_t *index, int32_t *population)
{ return cells_f64(N, pos, L, cutoff, ndiv, index, population); }

int32_t orc_cells_f32(int32_t N, const float *pos, double L, double cutoff, int32_t ndiv,
                      int32_t *index, int32_t *population)
{ return cells_f32(N, pos, L, cutoff, ndiv, index, population); }

void orc_nonbonded_cells_f64(int32_t N, const double *pos, double L, const orc_model64 *m, const orc_atom *atoms,
                             int nthreads, double *forces, double *energies, double *virials)
{ nonbonded_cells_f64(N, pos, L, m, atoms, nthreads, forces, energies, virials); }

void orc_nonbonded_cells_f32(int32_t N, const float *pos, float L, const orc_model32 *m, const orc_atom *atoms,
                             int nthreads, float *forces, float *energies, float *virials)
{ nonbonded_cells_f32(N, pos, L, m, atoms, nthreads, forces, energies, virials); }

static int cmp_i32(const void *a, const void *b)
{
    int32_t x = *(const int32_t *)a, y = *(const int32_t *)b;
    return (x > y) - (x < y);
}

int64_t orc_neighbor_list_f64(int32_t N, const double *pos, double L, double rlist,
                              int64_t *offsets, int32_t *nbrs)
{
    if (!(rlist <= 0.5 * L)) return -1;
    grid_f64 g;
    grid_build_f64(&g, N, pos, L, rlist);
    const int32_t M = g.M;
    const double rl2 = rlist * rlist;
    int64_t total = 0;
    offsets[0] = 0;
    for (int32_t i = 0; i < N; i++) {
        int32_t c = g.cell[i];
        int32_t cx = c % M, cy = (c / M) % M, cz = c / (M * M);
        int32_t sx[3], sy[3], sz[3];
        int nx = stencil1d_f64(cx, M, sx), ny = stencil1d_f64(cy, M, sy), nz = stencil1d_f64(cz, M, sz);
        int64_t first = total;
        for (int a = 0; a < nz; a++) for (int b = 0; b < ny; b++) for (int cc = 0; cc < nx; cc++) {
            int32_t nb = sx[cc] + M * (sy[b] + M * sz[a]);
            for (int32_t p = g.start[nb]; p < g.start[nb + 1]; p++) {
                int32_t j = g.order[p];
                if (j == i) continue;
                double r2 = 0.0;
                for (int d = 0; d < 3; d++) {
                    double rv = L * minimum_image_f64(pos[3 * i + d] / L - pos[3 * j + d] / L);
                    r2 += rv * rv;
                }
                if (r2 < rl2) { if (nbrs) nbrs[total] = j; total++; }
            }
        }
        if (nbrs) qsort(nbrs + first, (size_t)(total - first), sizeof(int32_t), cmp_i32);
        offsets[i + 1] = total;
    }
    grid_free_f64(&g);
    return total;
}

static void verlet_forces(int32_t N, const double *x, double L, const orc_model64 *m, const orc_atom *atoms,
                          int use_cells, int nthreads, double *f, double *e, double *w)
{
    if (use_cells) nonbonded_cells_f64(N, x, L, m, atoms, nthreads, f, e, w);
    else naive_f64(N, x, L, m, atoms, ORC_CUTOFF, f, e, w);
}

static void verlet_observe(int32_t N, const double *v, const double *inv_mass, const double *e, const double *w,
                           double *epot, double *ekin, double *vir)
{
    double se = 0.0, sw = 0.0, sk = 0.0;
    for (int32_t i = 0; i < N; i++) {
        se += e[i];
        sw += w[i];
        double mass = inv_mass ? 1.0 / inv_mass[i] : 1.0;
        sk += 0.5 * mass * (v[3 * i] * v[3 * i] + v[3 * i + 1] * v[3 * i + 1] + v[3 * i + 2] * v[3 * i + 2]);
    }
    if (epot) *epot = se;
    if (vir) *vir = sw;
    if (ekin) *ekin = sk;
}

void orc_verlet_f64(int32_t N, double *x, double *v, double L, const orc_model64 *m, const orc_atom *atoms,
                    const double *inv_mass, double dt, int32_t nsteps, int use_cells, int nthreads,
                    double *epot, double *ekin, double *virial, double *forces_out)
{
    double *f = (double *)malloc(sizeof(double) * 3 * (size_t)N);
    double *e = (double *)malloc(sizeof(double) * (size_t)N);
    double *w = (double *)malloc(sizeof(double) * (size_t)N);
    verlet_forces(N, x, L, m, atoms, use_cells, nthreads, f, e, w);
    verlet_observe(N, v, inv_mass, e, w, epot ? epot : NULL, ekin ? ekin : NULL, virial ? virial : NULL);
    for (int32_t s = 1; s <= nsteps; s++) {
        for (int32_t i = 0; i < N; i++) {
            double hdtm = 0.5 * dt * (inv_mass ? inv_mass[i] : 1.0);
            for (int d = 0; d < 3; d++) {
                v[3 * i + d] += hdtm * f[3 * i + d];
                x[3 * i + d] += dt * v[3 * i + d];
            }
        }
        verlet_forces(N, x, L, m, atoms, use_cells, nthreads, f, e, w);
        for (int32_t i = 0; i < N; i++) {
            double hdtm = 0.5 * dt * (inv_mass ? inv_mass[i] : 1.0);
            for (int d = 0; d < 3; d++) v[3 * i + d] += hdtm * f[3 * i + d];
        }
        verlet_observe(N, v, inv_mass, e, w, epot ? epot + s : NULL, ekin ? ekin + s : NULL,
                       virial ? virial + s : NULL);
    }
    if (forces_out) memcpy(forces_out, f, sizeof(double) * 3 * (size_t)N);
    free(f); free(e); free(w);
}

/* ---- Langevin thermostat (build-defined, like the integrator itself: SURVEY 8f item 4) -----------------
 * Counter-based normals: every (seed, step, atom id) gets its own three N(0,1) numbers, independent of the
 * order the atoms are stored or processed in.  mix = the splitmix64 finaliser. */
static uint64_t lgv_mix(uint64_t z)
{
    z ^= z >> 30; z *= 0xBF58476D1CE4E5B9ull;
    z ^= z >> 27; z *= 0x94D049BB133111EBull;
    z ^= z >> 31;
    return z;
}

void orc_langevin_normals(uint64_t seed, uint64_t step, uint64_t id, double out[3])
{
    const uint64_t base = lgv_mix(seed + 0x9E3779B97F4A7C15ull * id);
    const uint64_t s2 = lgv_mix(base ^ (0xD1B54A32D192ED03ull * (step + 1)));
    uint64_t r[4];
    for (int j = 0; j < 4; j++) r[j] = lgv_mix(s2 + 0x9E3779B97F4A7C15ull * (uint64_t)(j + 1));
    const double two53 = 1.0 / 9007199254740992.0, twopi = 6.283185307179586476925286766559;
    const double u1 = (double)((r[0] >> 11) + 1) * two53, v2 = (double)(r[1] >> 11) * two53;
    const double u3 = (double)((r[2] >> 11) + 1) * two53, v4 = (double)(r[3] >> 11) * two53;
    const double a = sqrt(-2.0 * log(u1)), b = sqrt(-2.0 * log(u3));
    out[0] = a * cos(twopi * v2);
    out[1] = a * sin(twopi * v2);
    out[2] = b * cos(twopi * v4);
}

/* velocity-Verlet with the O step between the opening half kick and the drift:
 *   v += (dt/2) f/m ; v = c1 v + c2 sqrt(T/m) xi(seed, step, id) ; x += dt v ; f = F(x) ; v += (dt/2) f/m
 * c1 = exp(-gamma dt), c2 = sqrt(1 - c1^2); steps are numbered step0, step0 + 1, ...; ids may be NULL (id = i). */
void orc_verlet_langevin_f64(int32_t N, double *x, double *v, double L, const orc_model64 *m, const orc_atom *atoms,
                             const double *inv_mass, double dt, int32_t nsteps, int use_cells, int nthreads,
                             double gamma, double temperature, uint64_t seed, uint64_t step0, const int64_t *ids,
                             double *epot, double *ekin, double *virial, double *forces_out)
{
    double *f = (double *)malloc(sizeof(double) * 3 * (size_t)N);
    double *e = (double *)malloc(sizeof(double) * (size_t)N);
    double *w = (double *)malloc(sizeof(double) * (size_t)N);
    const double c1 = exp(-gamma * dt), c2 = sqrt(1.0 - c1 * c1);
    verlet_forces(N, x, L, m, atoms, use_cells, nthreads, f, e, w);
    verlet_observe(N, v, inv_mass, e, w, epot ? epot : NULL, ekin ? ekin : NULL, virial ? virial : NULL);
    for (int32_t s = 1; s <= nsteps; s++) {
        for (int32_t i = 0; i < N; i++) {
            const double im = inv_mass ? inv_mass[i] : 1.0;
            const double hdtm = 0.5 * dt * im, amp = c2 * sqrt(temperature * im);
            double xi[3];
            orc_langevin_normals(seed, step0 + (uint64_t)(s - 1), (uint64_t)(ids ? ids[i] : i), xi);
            for (int d = 0; d < 3; d++) {
                double vv = v[3 * i + d] + hdtm * f[3 * i + d];
                vv = c1 * vv + amp * xi[d];
                v[3 * i + d] = vv;
                x[3 * i + d] += dt * vv;
            }
        }
        verlet_forces(N, x, L, m, atoms, use_cells, nthreads, f, e, w);
        for (int32_t i = 0; i < N; i++) {
            double hdtm = 0.5 * dt * (inv_mass ? inv_mass[i] : 1.0);
            for (int d = 0; d < 3; d++) v[3 * i + d] += hdtm * f[3 * i + d];
        }
        verlet_observe(N, v, inv_mass, e, w, epot ? epot + s : NULL, ekin ? ekin + s : NULL,
                       virial ? virial + s : NULL);
    }
    if (forces_out) memcpy(forces_out, f, sizeof(double) * 3 * (size_t)N);
    free(f); free(e); free(w);
}

int orc_max_threads(void) { return omp_get_max_threads(); }
