// impl_f64.hip -- fp64 instantiation (north-star precision) of every kernel and handle object.
#include "dd.hpp"

namespace emdee {
template struct Factory<double>;
}

namespace emdee {
void dd_rccl_selftest(emdee_ctx *ctx, int n_bytes) {
    EMDEE_REQUIRE(n_bytes >= 0, EMDEE_ERR_INVALID, "selftest: negative size");
    use_device(ctx);
    RcclApi &api = RcclApi::get();
    api.load();
    RcclApi::UniqueId id;
    EMDEE_RCCL_CHECK(api.GetUniqueId(&id));
    RcclApi::Comm comm = nullptr;
    EMDEE_RCCL_CHECK(api.CommInitRank(&comm, 1, id, 0));
    hipStream_t s = nullptr;
    EMDEE_HIP_CHECK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    DevBuf<unsigned char> a, b;
    DevBuf<double> r;
    const size_t n = (size_t)n_bytes;
    a.ensure(n + 16); b.ensure(n + 16); r.ensure(8);
    std::vector<unsigned char> h(n + 1), back(n + 1, 0);
    for (size_t i = 0; i < n; i++) h[i] = (unsigned char)((i * 131u + 7u) & 0xff);
    EMDEE_HIP_CHECK(hipMemcpyAsync(a.ptr, h.data(), n, hipMemcpyHostToDevice, s));
    EMDEE_HIP_CHECK(hipMemsetAsync(b.ptr, 0, n + 16, s));
    if (n > 0) {
        EMDEE_RCCL_CHECK(api.GroupStart());
        EMDEE_RCCL_CHECK(api.Send(a.ptr, n, RcclApi::kChar, 0, comm, s));
        EMDEE_RCCL_CHECK(api.Recv(b.ptr, n, RcclApi::kChar, 0, comm, s));
        EMDEE_RCCL_CHECK(api.GroupEnd());
    }
    const double v[3] = {1.5, -2.25, 1e300};
    double w[3] = {0, 0, 0};
    EMDEE_HIP_CHECK(hipMemcpyAsync(r.ptr, v, sizeof(v), hipMemcpyHostToDevice, s));
    EMDEE_RCCL_CHECK(api.AllReduce(r.ptr, r.ptr, 3, RcclApi::kFloat64, RcclApi::kSum, comm, s));
    EMDEE_HIP_CHECK(hipMemcpyAsync(w, r.ptr, sizeof(w), hipMemcpyDeviceToHost, s));
    EMDEE_HIP_CHECK(hipMemcpyAsync(back.data(), b.ptr, n, hipMemcpyDeviceToHost, s));
    EMDEE_HIP_CHECK(hipStreamSynchronize(s));
    (void)api.CommDestroy(comm);
    (void)hipStreamDestroy(s);
    EMDEE_REQUIRE(w[0] == v[0] && w[1] == v[1] && w[2] == v[2], EMDEE_ERR_HIP, "selftest: ncclAllReduce over one rank changed the values");
    for (size_t i = 0; i < n; i++)
        EMDEE_REQUIRE(back[i] == h[i], EMDEE_ERR_HIP, "selftest: byte %zu arrived as %u, sent %u", i, (unsigned)back[i], (unsigned)h[i]);
}
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

#ifdef EMDEE_BOUNDS
namespace emdee {
void bounds_poll_f64(int out[3]) { bounds_poll_here(out); }
}  // namespace emdee
#endif
