/* abi_client.c -- a plain C client of include/emdee_hip.h, the way a Julia `ccall` binding uses the library:
 * no Python, no torch, no HIP headers in this process; device memory and copies go through emdee_malloc /
 * emdee_memcpy_*.  It restates the reference's own test (test/runtests.jl:19-42: lj_sample.xyz, L = 10,
 * rc = 3, rs = 2.5, eps = sigma = 1, Float32 positions, bound 1e-4 between two implementations) with the
 * O(N) operator against the all-pairs operators, checks the fp64 path against the CPU oracle
 * (oracle/emdee_oracle.h: test infrastructure), and runs 50 velocity-Verlet steps against the oracle.
 * Usage: abi_client <lj_sample.xyz>.  Exit code 0 = all checks passed.  Built and run by
 * tests/test_gpu_c_client.py. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "emdee_hip.h"
#include "emdee_oracle.h"

#define CHECK(call)                                                                      \
    do {                                                                                 \
        int32_t rc_ = (call);                                                            \
        if (rc_ != EMDEE_OK) {                                                           \
            fprintf(stderr, "%s -> %d: %s\n", #call, (int)rc_, emdee_last_error());      \
            return 2;                                                                    \
        }                                                                                \
    } while (0)

static int failures = 0;
static void expect(int ok, const char *what, double got, double bound)
{
    printf("%-58s %.3e (bound %.1e) %s\n", what, got, bound, ok ? "ok" : "FAILED");
    if (!ok) failures++;
}

static double max_abs_diff_f(const float *a, const float *b, size_t n)
{
    double m = 0.0;
    for (size_t i = 0; i < n; i++) { double d = fabs((double)a[i] - (double)b[i]); if (d > m) m = d; }
    return m;
}
static double max_abs_diff_d(const double *a, const double *b, size_t n)
{
    double m = 0.0;
    for (size_t i = 0; i < n; i++) { double d = fabs(a[i] - b[i]); if (d > m) m = d; }
    return m;
}
static double max_abs_d(const double *a, size_t n)
{
    double m = 0.0;
    for (size_t i = 0; i < n; i++) if (fabs(a[i]) > m) m = fabs(a[i]);
    return m;
}

int main(int argc, char **argv)
{
    if (argc < 2) { fprintf(stderr, "usage: %s lj_sample.xyz\n", argv[0]); return 2; }
    FILE *fh = fopen(argv[1], "r");
    if (!fh) { perror(argv[1]); return 2; }
    int N = 0;
    char line[512];
    if (!fgets(line, sizeof line, fh) || sscanf(line, "%d", &N) != 1 || N <= 0) return 2;
    if (!fgets(line, sizeof line, fh)) return 2;                      /* comment line */
    double *x64 = malloc(sizeof(double) * 3 * N);
    float *x32 = malloc(sizeof(float) * 3 * N);
    for (int i = 0; i < N; i++) {
        char name[64];
        if (!fgets(line, sizeof line, fh) || sscanf(line, "%63s %lf %lf %lf", name, &x64[3 * i], &x64[3 * i + 1], &x64[3 * i + 2]) != 4) return 2;
        for (int d = 0; d < 3; d++) { x32[3 * i + d] = (float)x64[3 * i + d]; x64[3 * i + d] = (double)x32[3 * i + d]; }
    }
    fclose(fh);
    const double L = 10.0, rc = 3.0, rs = 2.5;
    const emdee_lj_model model = {rc * rc, rs * rs, 1.0 / (rc * rc - rs * rs)};
    emdee_lj_atom *atoms = malloc(sizeof(emdee_lj_atom) * N);
    for (int i = 0; i < N; i++) { atoms[i].half_sigma = 0.5f; atoms[i].twice_sqrt_eps = 2.0f; }   /* LennardJonesAtom(1, 1) */

    int32_t ndev = 0;
    CHECK(emdee_device_count(&ndev));
    emdee_ctx *ctx = NULL;
    CHECK(emdee_ctx_create(0, NULL, &ctx));
    char arch[64];
    int32_t cus = 0;
    int64_t hbm = 0;
    CHECK(emdee_device_info(ctx, arch, sizeof arch, &cus, &hbm));
    printf("libemdee_hip %d on %s, %d CUs, %.0f GB\n", (int)emdee_version(), arch, (int)cus, (double)hbm / 1e9);

    void *d_atoms, *d_x32, *d_x64, *d_f, *d_e, *d_w, *d_f2, *d_e2, *d_w2;
    CHECK(emdee_malloc(ctx, sizeof(emdee_lj_atom) * N, &d_atoms));
    CHECK(emdee_memcpy_h2d(ctx, d_atoms, atoms, sizeof(emdee_lj_atom) * N));
    CHECK(emdee_malloc(ctx, sizeof(float) * 3 * N, &d_x32));
    CHECK(emdee_memcpy_h2d(ctx, d_x32, x32, sizeof(float) * 3 * N));
    CHECK(emdee_malloc(ctx, sizeof(double) * 3 * N, &d_x64));
    CHECK(emdee_memcpy_h2d(ctx, d_x64, x64, sizeof(double) * 3 * N));
    CHECK(emdee_malloc(ctx, sizeof(double) * 3 * N, &d_f));  CHECK(emdee_malloc(ctx, sizeof(double) * N, &d_e));
    CHECK(emdee_malloc(ctx, sizeof(double) * N, &d_w));      CHECK(emdee_malloc(ctx, sizeof(double) * 3 * N, &d_f2));
    CHECK(emdee_malloc(ctx, sizeof(double) * N, &d_e2));     CHECK(emdee_malloc(ctx, sizeof(double) * N, &d_w2));

    /* ---- the reference's test, Float32: tiles operator vs the naive operator, LITERAL semantics ---- */
    float *f_a = malloc(sizeof(float) * 3 * N), *f_b = malloc(sizeof(float) * 3 * N), *e_a = malloc(sizeof(float) * N),
          *e_b = malloc(sizeof(float) * N), *w_a = malloc(sizeof(float) * N), *w_b = malloc(sizeof(float) * N);
    CHECK(emdee_compute_nonbonded_tiles(ctx, d_f, d_e, d_w, d_x32, L, N, model, d_atoms, 7, EMDEE_LITERAL, EMDEE_F32));
    CHECK(emdee_compute_nonbonded_naive(ctx, d_f2, d_e2, d_w2, d_x32, L, N, model, d_atoms, EMDEE_LITERAL, EMDEE_F32));
    CHECK(emdee_memcpy_d2h(ctx, f_a, d_f, sizeof(float) * 3 * N));  CHECK(emdee_memcpy_d2h(ctx, f_b, d_f2, sizeof(float) * 3 * N));
    CHECK(emdee_memcpy_d2h(ctx, e_a, d_e, sizeof(float) * N));      CHECK(emdee_memcpy_d2h(ctx, e_b, d_e2, sizeof(float) * N));
    CHECK(emdee_memcpy_d2h(ctx, w_a, d_w, sizeof(float) * N));      CHECK(emdee_memcpy_d2h(ctx, w_b, d_w2, sizeof(float) * N));
    double d = max_abs_diff_f(f_a, f_b, 3 * (size_t)N);
    expect(d < 1e-4, "f32 tiles vs naive operator, forces (runtests.jl:39)", d, 1e-4);
    d = max_abs_diff_f(e_a, e_b, N);
    expect(d < 1e-4, "f32 tiles vs naive operator, energies (runtests.jl:40)", d, 1e-4);
    d = max_abs_diff_f(w_a, w_b, N);
    expect(d < 1e-4, "f32 tiles vs naive operator, virials (runtests.jl:41)", d, 1e-4);

    /* ---- O(N) operator, fp64, against the CPU oracle (CUTOFF semantics) ---------------------------- */
    emdee_nbr *nbr = NULL;
    CHECK(emdee_nbr_create(ctx, N, 0.3, EMDEE_F64, &nbr));
    CHECK(emdee_compute_nonbonded(ctx, d_f, d_e, d_w, d_x64, L, nbr, model, d_atoms, 7, EMDEE_F64));
    double *f_g = malloc(sizeof(double) * 3 * N), *e_g = malloc(sizeof(double) * N), *w_g = malloc(sizeof(double) * N);
    double *f_o = malloc(sizeof(double) * 3 * N), *e_o = malloc(sizeof(double) * N), *w_o = malloc(sizeof(double) * N);
    CHECK(emdee_memcpy_d2h(ctx, f_g, d_f, sizeof(double) * 3 * N));
    CHECK(emdee_memcpy_d2h(ctx, e_g, d_e, sizeof(double) * N));
    CHECK(emdee_memcpy_d2h(ctx, w_g, d_w, sizeof(double) * N));
    const orc_model64 om = {model.rc2, model.rs2, model.inv_delta2};
    orc_naive_f64(N, x64, L, &om, (const orc_atom *)atoms, ORC_CUTOFF, f_o, e_o, w_o);
    d = max_abs_diff_d(f_g, f_o, 3 * (size_t)N) / max_abs_d(f_o, 3 * (size_t)N);
    expect(d < 1e-6, "f64 neighbour-list operator vs oracle, forces (relative)", d, 1e-6);
    d = max_abs_diff_d(e_g, e_o, N) / max_abs_d(e_o, N);
    expect(d < 1e-6, "f64 neighbour-list operator vs oracle, energies (relative)", d, 1e-6);
    d = max_abs_diff_d(w_g, w_o, N) / max_abs_d(w_o, N);
    expect(d < 1e-6, "f64 neighbour-list operator vs oracle, virials (relative)", d, 1e-6);
    int64_t builds = 0, listed = 0, pairs = 0;
    int32_t maxc = 0, cap = 0;
    CHECK(emdee_nbr_stats(nbr, &builds, &listed, &maxc, &cap));
    CHECK(emdee_nbr_count_pairs(nbr, &pairs));
    expect(pairs == 35677, "pairs with r < rc in lj_sample.xyz (SURVEY 8c: 35,677)", (double)pairs, 35677.0);

    /* ---- 50 velocity-Verlet steps from rest against the oracle ------------------------------------- */
    double *v0 = calloc(3 * (size_t)N, sizeof(double)), *xo = malloc(sizeof(double) * 3 * N), *vo = calloc(3 * (size_t)N, sizeof(double));
    memcpy(xo, x64, sizeof(double) * 3 * N);
    void *d_v;
    CHECK(emdee_malloc(ctx, sizeof(double) * 3 * N, &d_v));
    CHECK(emdee_memcpy_h2d(ctx, d_v, v0, sizeof(double) * 3 * N));
    const double lo[3] = {0, 0, 0}, len[3] = {L, L, L};
    const int32_t per[3] = {1, 1, 1};
    emdee_md *md = NULL;
    CHECK(emdee_md_create(ctx, lo, len, per, model, 0.3, EMDEE_F64, &md));
    CHECK(emdee_md_set_state(md, N, 0, d_x64, d_v, d_atoms, NULL));
    CHECK(emdee_md_step(md, 50, 0.002, 0));
    double tot[3], ep[51], ek[51];
    CHECK(emdee_md_energies(md, tot));
    CHECK(emdee_md_get_state(md, d_x64, d_v, NULL, NULL, NULL));
    double *xg = malloc(sizeof(double) * 3 * N);
    CHECK(emdee_memcpy_d2h(ctx, xg, d_x64, sizeof(double) * 3 * N));
    orc_verlet_f64(N, xo, vo, L, &om, (const orc_atom *)atoms, NULL, 0.002, 50, 0, 0, ep, ek, NULL, NULL);
    d = 0.0;
    for (size_t k = 0; k < 3 * (size_t)N; k++) {                     /* same trajectory up to the periodic image */
        double dx = xg[k] - xo[k];
        dx -= L * nearbyint(dx / L);
        if (fabs(dx) > d) d = fabs(dx);
    }
    expect(d < 1e-9, "50 velocity-Verlet steps vs oracle, positions", d, 1e-9);
    d = fabs(tot[0] / ep[50] - 1.0);
    expect(d < 1e-9, "potential energy after 50 steps (relative)", d, 1e-9);
    d = fabs(tot[1] / ek[50] - 1.0);
    expect(d < 1e-8, "kinetic energy after 50 steps (relative)", d, 1e-8);

    /* ---- error reporting across the boundary ------------------------------------------------------- */
    int32_t rcode = emdee_md_step(NULL, 1, 0.1, 0);
    expect(rcode != EMDEE_OK && strlen(emdee_last_error()) > 0, "NULL handle is an error code + message, not a crash", (double)rcode, 0.0);

    CHECK(emdee_md_destroy(md));
    CHECK(emdee_nbr_destroy(nbr));
    void *bufs[] = {d_atoms, d_x32, d_x64, d_f, d_e, d_w, d_f2, d_e2, d_w2, d_v};
    for (size_t k = 0; k < sizeof bufs / sizeof bufs[0]; k++) CHECK(emdee_free(ctx, bufs[k]));
    CHECK(emdee_ctx_destroy(ctx));
    printf("%s\n", failures ? "FAILED" : "all checks passed");
    return failures ? 1 : 0;
}
