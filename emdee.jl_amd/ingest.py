"""Inputs and outputs either side of the hot path: an XYZ reader/writer
(stands in for the Chemfiles read of test/runtests.jl:20-23), the Lennard-Jones part of an OpenMM-style
force-field file (the `NonbondedForce` table the reference parses at src/modelling.jl:71-73,197-200)
turned into LJAtom arrays, and a checkpoint of (positions, velocities, step).  Host-side, numpy only."""
import xml.etree.ElementTree as ET

import numpy as np

from .lennard_jones import lennard_jones_atoms


def read_xyz(path):
    """(names, positions (N, 3) float64) of the first frame of an XYZ file."""
    with open(path) as fh:
        n = int(fh.readline())
        fh.readline()
        names, pos = [], np.empty((n, 3), dtype=np.float64)
        for i in range(n):
            t = fh.readline().split()
            names.append(t[0])
            pos[i] = [float(t[1]), float(t[2]), float(t[3])]
    return names, pos


def write_xyz(path, names, positions, comment=""):
    positions = np.asarray(positions, dtype=np.float64)
    with open(path, "w") as fh:
        fh.write("%d\n%s\n" % (positions.shape[0], comment))
        for name, p in zip(names, positions):
            fh.write("%s %.12E %.12E %.12E\n" % (name, p[0], p[1], p[2]))


class NonbondedTable:
    """`<NonbondedForce lj14scale= coulomb14scale=><Atom type= sigma= epsilon= [charge=]/>...` -- the
    columns of the reference's NONBONDED frame (src/modelling.jl:71-73) and the two 1-4 scaling factors
    (:198-200, default 1.0).  sigma in nm and epsilon in kJ/mol as OpenMM writes them."""

    def __init__(self, xml_file):
        root = ET.parse(xml_file).getroot()
        nb = root.find("NonbondedForce")
        if nb is None:
            raise ValueError("no NonbondedForce element in %s" % xml_file)
        self.lj14scale = float(nb.attrib.get("lj14scale", 1.0))
        self.coulomb14scale = float(nb.attrib.get("coulomb14scale", 1.0))
        self.types = {}
        for atom in nb.findall("Atom"):
            self.types[atom.attrib["type"]] = dict(sigma=float(atom.attrib["sigma"]), epsilon=float(atom.attrib["epsilon"]),
                                                   charge=float(atom.attrib.get("charge", 0.0)))

    def lj_atoms(self, atom_types, length_unit=1.0, energy_unit=1.0):
        """LJAtom array for a sequence of type names; sigma / length_unit and epsilon / energy_unit convert to
        the caller's units (e.g. length_unit = 0.1 for Angstrom positions with nm force-field sigmas)."""
        sigma = np.array([self.types[t]["sigma"] for t in atom_types], dtype=np.float64) / length_unit
        eps = np.array([self.types[t]["epsilon"] for t in atom_types], dtype=np.float64) / energy_unit
        return lennard_jones_atoms(eps, sigma)


def save_checkpoint(path, positions, velocities, step, box_length):
    """(x, v, step, L) as a compressed npz; accepts GPU tensors or numpy arrays."""
    to_np = lambda a: a.detach().cpu().numpy() if hasattr(a, "detach") else np.asarray(a)
    np.savez_compressed(path, positions=to_np(positions), velocities=to_np(velocities), step=int(step), L=float(box_length))


def load_checkpoint(path):
    d = np.load(path)
    return d["positions"], d["velocities"], int(d["step"]), float(d["L"])
