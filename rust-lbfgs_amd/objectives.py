"""Device-resident objectives: the `evaluate` closure kept in HBM (lbfgs_hip_objective).

`Quadratic` and `Logistic` are the synthetic workloads of BASELINE.json configs 2-4; their
data is a counter-based hash of the GLOBAL element index (nothing stored, identical on every
rank and on the CPU oracle).  `Rosenbrock` is the reference's default_evaluate (src/lib.rs:79-94).
"""
from . import _ffi
from .api import BuiltinObjective

SEED_QUAD_A, SEED_QUAD_B = 0x5EED0001, 0x5EED0002
SEED_LOGI_A, SEED_LOGI_T = 0x5EED0003, 0x5EED0004


# fuse_line_eval (lbfgs_solver.h): 0/False = separate passes per trial, 1/True = one pass per trial,
# 2 = trials write no vectors, the accepted point is formed in the history update's pass (element-wise objectives)
def Quadratic(fuse_line_eval=2):
    """f = sum x_i*(0.5*a_i*x_i - b_i), a_i = 1 + 999*u_i^2 (cond 1e3), b_i = 2*u'_i - 1."""
    return BuiltinObjective(_ffi.OBJ_QUADRATIC, SEED_QUAD_A, SEED_QUAD_B, fuse_line_eval)


def Logistic(fuse_line_eval=2):
    """f = sum log(1+exp(-t_i*a_i*x_i)), a_i = 0.5 + 1.5*u_i, t_i = +-1."""
    return BuiltinObjective(_ffi.OBJ_LOGISTIC, SEED_LOGI_A, SEED_LOGI_T, fuse_line_eval)


def Rosenbrock(fuse_line_eval=True):
    return BuiltinObjective(_ffi.OBJ_ROSENBROCK, 0, 0, fuse_line_eval)


def LennardJones():
    """examples/lj.rs:20-64,113-118: exact all-pairs LJ (epsilon = sigma = 1); x = 3*natoms coordinates; one rank."""
    return BuiltinObjective(_ffi.OBJ_LJ_ALLPAIRS, fuse_line_eval=True)


def LennardJonesNeighbors(nbr_index, cutoff):
    """Substitute evaluator for BASELINE config 5 at scale: the same pair terms over a fixed neighbour table
    (int32 [max_nbr, natoms], -1 = empty, every pair listed from both ends) with a cutoff, energy shifted by v(rc)."""
    return BuiltinObjective(_ffi.OBJ_LJ_NEIGHBORS, fuse_line_eval=True, nbr_index=nbr_index, cutoff=float(cutoff))


def LennardJonesCells(cutoff=2.5, skin=0.3, max_nbr=0):
    """The evaluator of BASELINE config 5 at scale: E = sum_{i<j, r<rc} [v(r) - v(rc)] through a neighbour list the
    library builds ON THE DEVICE from a cell list (radius cutoff + skin) and rebuilds whenever an atom has moved more
    than skin/2 since the last build, so atoms may travel freely during a minimisation.  Deviation from
    examples/lj.rs:38-64 (documented): the cutoff and the energy shift; the pair terms are the example's."""
    return BuiltinObjective(_ffi.OBJ_LJ_CELLS, fuse_line_eval=True, cutoff=float(cutoff), skin=float(skin),
                            max_nbr=int(max_nbr))


def cubic_lattice_neighbors(nside, spacing, cutoff):
    """Neighbour table of an nside^3 simple-cubic lattice (open boundaries): every site within `cutoff` at the
    ideal positions, so it stays valid while atoms move by less than half the skin.  -> (x0 [3*natoms], table)."""
    import numpy as np

    r = int(np.floor(cutoff / spacing))
    offs = [(a, b, c) for a in range(-r, r + 1) for b in range(-r, r + 1) for c in range(-r, r + 1)
            if (a, b, c) != (0, 0, 0) and (a * a + b * b + c * c) * spacing * spacing < cutoff * cutoff]
    n = nside ** 3
    idx = np.arange(n, dtype=np.int64)
    ix, iy, iz = idx // (nside * nside), (idx // nside) % nside, idx % nside
    tab = np.full((len(offs), n), -1, dtype=np.int32)
    for k, (a, b, c) in enumerate(offs):
        jx, jy, jz = ix + a, iy + b, iz + c
        ok = (jx >= 0) & (jx < nside) & (jy >= 0) & (jy < nside) & (jz >= 0) & (jz < nside)
        tab[k, ok] = ((jx * nside + jy) * nside + jz)[ok].astype(np.int32)
    x0 = np.stack([ix, iy, iz], axis=1).astype(np.float64).reshape(-1) * spacing
    return x0, tab
