"""LennardJonesModel / LennardJonesAtom / interaction -- host mirror of src/lennard_jones.jl."""
import ctypes as C
import math

import numpy as np
import torch

from . import _lib
from .device import context_for, precision_of

# LJAtom{half_σ::Float32, twice_sqrt_ε::Float32} -- src/lennard_jones.jl:15-18
LJAtom = np.dtype([("half_sigma", np.float32), ("twice_sqrt_eps", np.float32)])


class LennardJonesModel:
    """LennardJonesModel(cutoff, switch) -- src/lennard_jones.jl:6-11.

    Fields rc2, rs2, inv_delta2 = cutoff^2, switch^2, 1/(cutoff^2 - switch^2).  The reference stores
    them as Float32; they are kept here in double and rounded to Float32 by the fp32 kernels, which is
    the same value.  switch == cutoff raises (the reference would build an Inf).
    """

    def __init__(self, cutoff, switch):
        cutoff, switch = float(cutoff), float(switch)
        if not (0.0 <= switch < cutoff):
            raise ValueError("LennardJonesModel needs 0 <= switch < cutoff")
        self.cutoff, self.switch = cutoff, switch
        self.rc2 = cutoff ** 2
        self.rs2 = switch ** 2
        self.inv_delta2 = 1.0 / (cutoff ** 2 - switch ** 2)

    def __repr__(self):
        return "LennardJonesModel(cutoff=%g, switch=%g)" % (self.cutoff, self.switch)


def LennardJonesAtom(eps, sigma):
    """LennardJonesAtom(ε, σ) = LJAtom(0.5σ, 2*sqrt(ε)) -- src/lennard_jones.jl:13."""
    a = np.zeros((), dtype=LJAtom)
    a["half_sigma"] = np.float32(0.5 * float(sigma))
    a["twice_sqrt_eps"] = np.float32(2.0 * math.sqrt(float(eps)))
    return a


def lennard_jones_atoms(eps, sigma, n=None):
    """Vectorised LennardJonesAtom: arrays (or scalars broadcast to n) -> LJAtom array."""
    eps = np.atleast_1d(np.asarray(eps, dtype=np.float64))
    sigma = np.atleast_1d(np.asarray(sigma, dtype=np.float64))
    if n is not None:
        eps, sigma = np.broadcast_to(eps, (n,)), np.broadcast_to(sigma, (n,))
    out = np.empty(eps.shape[0], dtype=LJAtom)
    out["half_sigma"] = (0.5 * sigma).astype(np.float32)
    out["twice_sqrt_eps"] = (2.0 * np.sqrt(eps)).astype(np.float32)
    return out


def _atom_c(a):
    a = np.asarray(a, dtype=LJAtom).reshape(())
    return _lib.LJAtomC(float(a["half_sigma"]), float(a["twice_sqrt_eps"]))


def interaction(r2, model, atom_i, atom_j, mode=_lib.LITERAL):
    """interaction(r², model, atom_i, atom_j) -> (E, minus_E′r) -- src/lennard_jones.jl:25-42,
    evaluated by the device pair function on a tensor of r² values (float32 or float64, on the GPU)."""
    if not (isinstance(r2, torch.Tensor) and r2.is_cuda):
        raise TypeError("interaction() runs on the GPU: r2 must be a CUDA/HIP tensor")
    r2 = r2.contiguous()
    E, W = torch.empty_like(r2), torch.empty_like(r2)
    ctx = context_for(r2.device)
    _lib.call("emdee_interaction", ctx.handle, r2.numel(), r2.data_ptr(), _lib.model_c(model), _atom_c(atom_i),
              _atom_c(atom_j), int(mode), E.data_ptr(), W.data_ptr(), precision_of(r2))
    return E, W
