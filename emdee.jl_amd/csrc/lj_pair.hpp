// lj_pair.hpp -- the device pair function.
// Restates interaction(r2, model, atom_i, atom_j) of the reference, src/lennard_jones.jl:25-42.
// Same function, fewer fp64 instructions (this kernel is fp64-issue bound on gfx950, 4 cycles per
// wave64 instruction): one reciprocal of r2 shared by s^-2 = sigma^2/r2 (:31) and by the caller's
// W/r2 (src/nonbonded.jl:74,139); E and W/6 from one product t = 4 eps s^-12 (E = t - 4 eps s^-6,
// W/6 = E + t); the switch polynomials in Horner/product form.  Differences from the literal
// operation order are a few ulp.
#pragma once

#include <hip/hip_runtime.h>

namespace emdee {

// LennardJonesModel, src/lennard_jones.jl:6-11, in the kernel's real type, plus constants derived
// once on the host: x = r2 idl2 - rs2 idl2 ;  -r g' = 60 idl2 x^2 (1-x)^2 r2
template <typename real>
struct LJModel {
    real rc2, rs2, idl2;
    real x0;      // rs2 * idl2
    real c60;     // 60 * idl2
    real k3, k6;  // 3 and 6 as kernel arguments: SGPR operands instead of per-iteration literal moves
    real k18, k36;  // 6 x the switch polynomial 1 + 3 x + 6 x^2 (force-only kernels fold W's factor 6 into it)
    real nx0;       // -x0
    real h3, h4, h5;  // 6 g = 6 + x^3 (h3 + h4 x + h5 x^2) = 6 - 60 x^3 + 90 x^4 - 36 x^5 (force-only kernels, Horner form)
};

template <typename real>
static inline LJModel<real> make_model(const emdee_lj_model &m) {
    // EMDEE_F32: round each field to float = the reference's Float32 struct
    LJModel<real> r;
    r.rc2 = (real)m.rc2; r.rs2 = (real)m.rs2; r.idl2 = (real)m.inv_delta2;
    r.x0 = r.rs2 * r.idl2;
    r.c60 = (real)60 * r.idl2;
    r.k3 = (real)3; r.k6 = (real)6; r.k18 = (real)18; r.k36 = (real)36;
    r.nx0 = -r.x0; r.h3 = (real)-60; r.h4 = (real)90; r.h5 = (real)-36;
    return r;
}

__device__ __forceinline__ float fast_rcp(float a) { return __builtin_amdgcn_rcpf(a); }   // v_rcp_f32, 1 ulp
__device__ __forceinline__ double fast_rcp(double a) {
    // v_rcp_f64 seed (measured on gfx950: 4.5e-8 relative) + one Newton step -> 2.2e-15 relative
    // (profiles/README.md), without the ~12-instruction IEEE division sequence
    double r = __builtin_amdgcn_rcp(a);
    r = __builtin_fma(__builtin_fma(-a, r, 1.0), r, r);
    return r;
}

// The reference clamp  x *= 0.5 (sign(x) - sign(x-1))  (src/lennard_jones.jl:37):
// 0 < x < 1 -> x ;  x <= 0 -> 0 ;  x == 1 -> 0.5 (Q2) ;  x > 1 -> 0 (g = 1 beyond rc, Q1).
// One max covers the common cases; x >= 1 is a rare, separately handled lane state.
template <typename real>
__device__ __forceinline__ real switch_clamp(real x) {
    x = x > (real)0 ? x : (real)0;
    if (__builtin_expect(x >= (real)1, 0)) x = (x == (real)1) ? (real)0.5 : (real)0;
    return x;
}

// Returns (E g, W g + E (-r g')) given r2, inv_r2 = 1/r2 and the pair constants sigma2 = sigma_ij^2,
// e4 = 4 eps_ij.  CUTOFF semantics (r2 >= rc2 contributes nothing) are the CALLER's test; this is the
// literal function.
template <typename real>
__device__ __forceinline__ void lj_interaction_pair(real r2, real inv_r2, const LJModel<real> &m, real sigma2, real e4,
                                                    real &E_out, real &W_out) {
    const real s2 = sigma2 * inv_r2;                                   // :31
    const real s6 = s2 * s2 * s2;                                      // :32
    const real e4s6 = e4 * s6;                                         // :33  4 eps s^-6
    const real t = e4s6 * s6;                                          //      4 eps s^-12
    const real E = t - e4s6;                                           // :34  4 eps (s^-12 - s^-6)
    const real W = (real)6 * (E + t);                                  // :35  24 eps (2 s^-12 - s^-6)
    const real x = switch_clamp(r2 * m.idl2 - m.x0);                   // :36-37
    const real x2 = x * x;                                             // :38
    // :39  g = 1 + x^3 (15 x - 6 x^2 - 10) = (1-x)^3 (1 + 3x + 6x^2)   (same quintic, shares u = 1-x with g')
    const real u = (real)1 - x;
    const real u2 = u * u;
    const real g = (u2 * u) * ((real)1 + x * (m.k3 + m.k6 * x));
    const real mgr = m.c60 * (x2 * u2) * r2;                           // :40  -r g' = 60 x^2 (1-x)^2 idl2 r2
    E_out = E * g;                                                     // :41
    W_out = W * g + E * mgr;
}

// x = clamp(a b + c, 0, 1) in one instruction (the VOP3 clamp bit)
__device__ __forceinline__ double fma_clamp01(double a, double b, double c) {
    double r;
    asm("v_fma_f64 %0, %1, %2, %3 clamp" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
__device__ __forceinline__ float fma_clamp01(float a, float b, float c) {
    float r;
    asm("v_fma_f32 %0, %1, %2, %3 clamp" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
// Force-only launches need neither E nor W on their own, only (W g + E (-r g')) / r2.  Two products fall out of the
// algebra: the factor 6 of W = 6 (2 b - a) (a = 4 eps s^-6, b = a s^-6) moves into the constants of the switch
// polynomial, and -r g' / r2 = 60 idl2 x^2 (1-x)^2 needs no r2 at all.  Same function as lj_interaction_pair
// followed by (W g + E mgr) * inv_r2, two instructions shorter, different only in rounding.
template <typename real>
__device__ __forceinline__ real lj_force_over_r2(real r2, real inv_r2, const LJModel<real> &m, real sigma2, real e4) {
    const real s2 = sigma2 * inv_r2;
    const real s6 = s2 * s2 * s2;
    const real a = e4 * s6;                                            // 4 eps s^-6
    const real b = a * s6;                                             // 4 eps s^-12
    const real d = (real)2 * b - a;                                    // W / 6
    const real em = b - a;                                             // E
    // x clamped to [0, 1] by the instruction itself; under the caller's test r2 < rc2 only a sum that ROUNDS to exactly
    // 1.0 differs from the literal clamp (x == 1 -> 0.5, Q2): it yields 0 here, the limit of the function at the cutoff
    const real x = fma_clamp01(r2, m.idl2, m.nx0);
    const real x2 = x * x;
    real t = m.h5 * x + m.h4;
    t = t * x + m.h3;
    const real g6 = (x2 * x) * t + m.k6;                               // 6 g = 6 - 60 x^3 + 90 x^4 - 36 x^5
    const real y = x - x2;                                             // x (1 - x)
    const real q = (m.c60 * y) * y;                                    // -r g' / r2
    return (d * g6) * inv_r2 + em * q;
}

// ---- two-species (typed) MD kernels, force-only launches: 4 eps_ij of the segment folded into the switch constants ------
// lj_force_over_r2 with a = s^-6 instead of 4 eps s^-6: the factor moves into 6 g's Horner constants and into 60 idl2, which a
// lane of the typed kernels holds per neighbour species anyway (csrc/typed.hpp).  One multiply per pair step less; different
// from lj_force_over_r2 only in rounding.
template <typename real>
struct LJSeg {
    real sig2;                 // sigma_ij^2
    real p0, p3, p4, p5;       // 4 eps_ij (6, -60, 90, -36)
    real c60;                  // 4 eps_ij 60 idl2
};
template <typename real>
__host__ __device__ inline LJSeg<real> make_seg(const LJModel<real> &m, real sig2, real e4) {
    LJSeg<real> c;
    c.sig2 = sig2;
    c.p0 = m.k6 * e4; c.p3 = m.h3 * e4; c.p4 = m.h4 * e4; c.p5 = m.h5 * e4;
    c.c60 = m.c60 * e4;
    return c;
}
template <typename real>
__device__ __forceinline__ real lj_force_over_r2_seg(real r2, real inv_r2, const LJModel<real> &m, const LJSeg<real> &c) {
    const real s2 = c.sig2 * inv_r2;
    const real s6 = s2 * s2 * s2;                                      // a / 4 eps
    const real b = s6 * s6;                                            // b / 4 eps
    const real d = (real)2 * b - s6;                                   // W / (6 . 4 eps)
    const real em = b - s6;                                            // E / 4 eps
    const real x = fma_clamp01(r2, m.idl2, m.nx0);
    const real x2 = x * x;
    real t = c.p5 * x + c.p4;
    t = t * x + c.p3;
    const real g6 = (x2 * x) * t + c.p0;                               // 4 eps 6 g
    const real y = x - x2;
    const real q = (c.c60 * y) * y;                                    // 4 eps (-r g') / r2
    return (d * g6) * inv_r2 + em * q;
}

// ---- single-species MD kernels: the same function with everything constant folded into the launch constants ------
// All atoms carry one LJAtom, so sigma and 4 eps are launch constants.  The kernel works in coordinates scaled by
// 1/sigma (done once per record while the tile is staged): s^-2 is then 1/r'^2 itself; 4 eps is folded into the
// constants of the switch polynomials; 6 g = 6 - 60 x^3 + 90 x^4 - 36 x^5 in Horner form and -r g'/r^2 = 60 idl2
// (x - x^2)^2 share nothing but x; and x = clamp(r'^2 idl2' - x0, 0, 1) is ONE instruction (the VOP3 clamp bit).
// The caller's test r'^2 < rc'^2 keeps x < 1 except when the fused multiply-add rounds to exactly 1.0, where this
// form yields 0 (the limit of the function as r -> rc) and not the reference's x == 1 -> 0.5 quirk (Q2): the
// all-pairs kernels and every kernel that outputs E or W keep the literal clamp.
// Returns (W g + E (-r g')) / r'^2 in units where the caller still owes one factor 1/sigma on the summed force.
template <typename real>
struct LJUni {
    real rc2;          // rc^2 / sigma^2
    real idl2;         // idl2 sigma^2
    real nx0;          // -(rs2 idl2)
    real p0, p3, p4, p5;   // 4 eps (6, -60, 90, -36)
    real c60;          // 4 eps 60 idl2 sigma^2
    real inv_sigma;    // 1 / sigma
};
template <typename real>
static inline LJUni<real> make_uni(const LJModel<real> &m, real sigma, real e4) {
    LJUni<real> u;
    const real s2 = sigma * sigma;
    u.rc2 = m.rc2 / s2;
    u.idl2 = m.idl2 * s2;
    u.nx0 = -m.x0;
    u.p0 = (real)6 * e4; u.p3 = (real)-60 * e4; u.p4 = (real)90 * e4; u.p5 = (real)-36 * e4;
    u.c60 = e4 * (real)60 * u.idl2;
    u.inv_sigma = (real)1 / sigma;
    return u;
}
template <typename real>
__device__ __forceinline__ real lj_force_over_r2_uni(real r2, const LJUni<real> &m) {
    const real x = fma_clamp01(r2, m.idl2, m.nx0);
    const real inv = fast_rcp(r2);                                     // s^-2 in scaled coordinates
    const real s6 = inv * inv * inv;                                   // a / 4 eps
    const real b = s6 * s6;                                            // b / 4 eps
    const real d = (real)2 * b - s6;                                   // W / (6 . 4 eps)   (contracted to one fma)
    const real em = b - s6;                                            // E / 4 eps
    const real x2 = x * x;
    real t = m.p5 * x + m.p4;
    t = t * x + m.p3;
    const real g6 = (x2 * x) * t + m.p0;                               // 4 eps 6 g
    const real y = x - x * x;                                          // x (1 - x)
    const real q = (m.c60 * y) * y;                                    // 4 eps (-r g') / r'^2
    return em * q + (d * g6) * inv;
}

// ---- two pairs per lane in packed fp32 (plain fp32 VALU instructions issue at the fp64 rate on gfx950) ------
typedef float f32x2 __attribute__((ext_vector_type(2)));

// the same function as lj_interaction_pair, on two independent pairs at once
__device__ __forceinline__ void lj_interaction_pair2(f32x2 r2, f32x2 inv_r2, const LJModel<float> &m, f32x2 sigma2, f32x2 e4,
                                                     f32x2 &E_out, f32x2 &W_out) {
    const f32x2 s2 = sigma2 * inv_r2;
    const f32x2 s6 = s2 * s2 * s2;
    const f32x2 e4s6 = e4 * s6;
    const f32x2 t = e4s6 * s6;
    const f32x2 E = t - e4s6;
    const f32x2 W = 6.0f * (E + t);
    f32x2 x = r2 * m.idl2 - m.x0;
    x.x = switch_clamp(x.x);
    x.y = switch_clamp(x.y);
    const f32x2 x2 = x * x;
    const f32x2 u = 1.0f - x;
    const f32x2 u2 = u * u;
    const f32x2 g = (u2 * u) * (1.0f + x * (m.k3 + m.k6 * x));
    const f32x2 mgr = m.c60 * (x2 * u2) * r2;
    E_out = E * g;
    W_out = W * g + E * mgr;
}
// lj_force_over_r2 on two independent pairs at once
__device__ __forceinline__ f32x2 lj_force_over_r2_2(f32x2 r2, f32x2 inv_r2, const LJModel<float> &m, f32x2 sigma2, f32x2 e4) {
    const f32x2 s2 = sigma2 * inv_r2;
    const f32x2 s6 = s2 * s2 * s2;
    const f32x2 a = e4 * s6;
    const f32x2 b = a * s6;
    const f32x2 d = 2.0f * b - a;
    const f32x2 em = b - a;
    f32x2 x = r2 * m.idl2 - m.x0;
    x.x = switch_clamp(x.x);
    x.y = switch_clamp(x.y);
    const f32x2 x2 = x * x;
    const f32x2 u = 1.0f - x;
    const f32x2 u2 = u * u;
    const f32x2 g6 = (u2 * u) * (m.k6 + x * (m.k18 + m.k36 * x));
    const f32x2 q = m.c60 * (x2 * u2);
    return (d * g6) * inv_r2 + em * q;
}
// lj_force_over_r2_uni on two independent pairs at once (scaled coordinates, constants folded; see LJUni)
__device__ __forceinline__ f32x2 pk_fma_clamp01(f32x2 a, f32x2 b, f32x2 c) {
    f32x2 r;
    asm("v_pk_fma_f32 %0, %1, %2, %3 clamp" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
__device__ __forceinline__ f32x2 lj_force_over_r2_uni2(f32x2 r2, const LJUni<float> &m) {
    const f32x2 x = pk_fma_clamp01(r2, f32x2{m.idl2, m.idl2}, f32x2{m.nx0, m.nx0});
    const f32x2 inv = {fast_rcp(r2.x), fast_rcp(r2.y)};
    const f32x2 s6 = inv * inv * inv;
    const f32x2 b = s6 * s6;
    const f32x2 d = 2.0f * b - s6;
    const f32x2 em = b - s6;
    const f32x2 x2 = x * x;
    f32x2 t = m.p5 * x + m.p4;
    t = t * x + m.p3;
    const f32x2 g6 = (x2 * x) * t + m.p0;
    const f32x2 y = x - x * x;
    const f32x2 q = (m.c60 * y) * y;
    return em * q + (d * g6) * inv;
}
__device__ __forceinline__ f32x2 lj_force_over_r2_uni2(f32x2, const LJUni<double> &) { return f32x2{0.f, 0.f}; }
__device__ __forceinline__ f32x2 lj_force_over_r2_2(f32x2, f32x2, const LJModel<double> &, f32x2, f32x2) { return f32x2{0.f, 0.f}; }
// (double instantiations never call it; the overload keeps `if constexpr` branches well-formed)
__device__ __forceinline__ void lj_interaction_pair2(f32x2, f32x2, const LJModel<double> &, f32x2, f32x2, f32x2 &, f32x2 &) {}

// Lorentz-Berthelot through the LJAtom encoding: sigma_ij = half_sigma_i + half_sigma_j (:29),
// 4 eps_ij = twice_sqrt_eps_i * twice_sqrt_eps_j (:30,33)
template <typename real>
__device__ __forceinline__ void lj_interaction(real r2, real inv_r2, const LJModel<real> &m, real hs_i, real te_i,
                                               real hs_j, real te_j, real &E_out, real &W_out) {
    const real sigma = hs_i + hs_j;
    lj_interaction_pair(r2, inv_r2, m, sigma * sigma, te_i * te_j, E_out, W_out);
}

}  // namespace emdee
