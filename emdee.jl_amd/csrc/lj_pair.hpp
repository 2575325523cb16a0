// lj_pair.hpp -- the device pair function.
// Follows interaction(r2, model, atom_i, atom_j) of the reference, src/lennard_jones.jl:25-42,
// term by term; the only algebraic change is one reciprocal of r2 shared by s^-2 = sigma^2/r2
// (:31) and by the caller's W/r2 (src/nonbonded.jl:74,139).
#pragma once

#include <hip/hip_runtime.h>

namespace emdee {

// LennardJonesModel, src/lennard_jones.jl:6-11, in the kernel's real type
template <typename real>
struct LJModel {
    real rc2, rs2, idl2;
};

template <typename real>
static inline LJModel<real> make_model(const emdee_lj_model &m) {
    // EMDEE_F32: round each field to float = the reference's Float32 struct
    return LJModel<real>{(real)m.rc2, (real)m.rs2, (real)m.inv_delta2};
}

__device__ __forceinline__ float fast_rcp(float a) { return __builtin_amdgcn_rcpf(a); }   // v_rcp_f32, 1 ulp
__device__ __forceinline__ double fast_rcp(double a) {
    // v_rcp_f64 seed + two Newton steps: full fp64 accuracy without the IEEE division sequence
    double r = __builtin_amdgcn_rcp(a);
    r = __builtin_fma(__builtin_fma(-a, r, 1.0), r, r);
    r = __builtin_fma(__builtin_fma(-a, r, 1.0), r, r);
    return r;
}

// The reference clamp  x *= 0.5 (sign(x) - sign(x-1))  (src/lennard_jones.jl:37) as selects:
// 0 < x < 1 -> x ;  x == 1 -> 0.5 (Q2) ;  otherwise 0 (including x > 1: g = 1 beyond rc, Q1).
template <typename real>
__device__ __forceinline__ real switch_clamp(real x) {
    real inside = (x > (real)0 && x < (real)1) ? x : (real)0;
    return (x == (real)1) ? (real)0.5 : inside;
}

// Returns (E g, W g + E (-r g')) given r2 and inv_r2 = 1/r2.  CUTOFF semantics (r2 >= rc2
// contributes nothing) are the CALLER's test; this is the literal formula.
template <typename real>
__device__ __forceinline__ void lj_interaction(real r2, real inv_r2, const LJModel<real> &m, real hs_i, real te_i,
                                               real hs_j, real te_j, real &E_out, real &W_out) {
    real sigma = hs_i + hs_j;                                          // :29
    real s2 = sigma * sigma * inv_r2;                                  // :31
    real s6 = s2 * s2 * s2;                                            // :32
    real e4s6 = te_i * te_j * s6;                                      // :33
    real E = e4s6 * (s6 - (real)1);                                    // :34
    real W = (real)6 * e4s6 * ((real)2 * s6 - (real)1);                // :35
    real x = switch_clamp((r2 - m.rs2) * m.idl2);                      // :36-37
    real x2 = x * x;                                                   // :38
    real g = (real)1 + x * x2 * ((real)15 * x - (real)6 * x2 - (real)10);            // :39
    real mgr = (real)60 * x2 * ((real)1 - (real)2 * x + x2) * m.idl2 * r2;            // :40
    E_out = E * g;                                                     // :41
    W_out = W * g + E * mgr;
}

}  // namespace emdee
