// impl_f64.hip -- fp64 instantiation (north-star precision) of every kernel and handle object.
#include "dd.hpp"

namespace emdee {
template struct Factory<double>;
}

namespace emdee {
void dd_describe(const double len[3], const int32_t grid[3], double halo, int rank, int32_t *ndirs, int32_t *dirs,
                 int32_t *dir_rank, double *dir_shift, int32_t *npeers, int32_t *peers, double *local_lo, double *local_len,
                 int32_t *periodic) {
    const int g[3] = {grid[0], grid[1], grid[2]};
    EMDEE_REQUIRE(g[0] >= 1 && g[1] >= 1 && g[2] >= 1, EMDEE_ERR_INVALID, "emdee_dd_describe: grid must be positive");
    EMDEE_REQUIRE(rank >= 0 && rank < g[0] * g[1] * g[2], EMDEE_ERR_INVALID, "emdee_dd_describe: rank outside the grid");
    DdGeom geo;
    geo.init(len, g, halo, rank);
    *ndirs = geo.ndirs;
    for (int k = 0; k < geo.ndirs; k++) {
        for (int d = 0; d < 3; d++) { dirs[3 * k + d] = geo.dir[k][d]; dir_shift[3 * k + d] = geo.dir_shift[k][d]; }
        dir_rank[k] = geo.dir_rank[k];
    }
    *npeers = geo.npeers;
    for (int p = 0; p < geo.npeers; p++) peers[p] = geo.peers[p];
    for (int d = 0; d < 3; d++) { local_lo[d] = geo.local_lo[d]; local_len[d] = geo.local_len[d]; periodic[d] = geo.periodic[d]; }
}
}  // namespace emdee
