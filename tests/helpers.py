"""Shared test helpers: pyref (canonical python ints) <-> C-ABI numpy limb arrays (Montgomery)."""
import numpy as np
import pyref as P
import cref

M64 = 0xFFFFFFFFFFFFFFFF


def fr_arr(vals):
    return np.array([cref.int_to_limbs(P.fr_to_mont(v % P.R_MOD)) for v in vals], dtype=np.uint64).reshape(-1, 4)


def fr_vals(arr):
    return [P.fr_from_mont(cref.limbs_to_int(row)) for row in np.asarray(arr).reshape(-1, 4)]


def fp_arr(vals):
    return np.array([cref.int_to_limbs(P.fp_to_mont(v % P.Q_MOD)) for v in vals], dtype=np.uint64).reshape(-1, 4)


def fp_vals(arr):
    return [P.fp_from_mont(cref.limbs_to_int(row)) for row in np.asarray(arr).reshape(-1, 4)]


def g1_arr(pts):
    rows = []
    for pt in pts:
        rows.append([0] * 8 if pt is None else cref.int_to_limbs(P.fp_to_mont(pt[0])) + cref.int_to_limbs(P.fp_to_mont(pt[1])))
    return np.array(rows, dtype=np.uint64).reshape(-1, 8)


def g1_pts(arr):
    out = []
    for row in np.asarray(arr).reshape(-1, 8):
        v = fp_vals(row.reshape(2, 4))
        out.append(None if v == [0, 0] else (v[0], v[1]))
    return out


def g2_arr(pts):
    rows = []
    for pt in pts:
        if pt is None:
            rows.append([0] * 16)
        else:
            (x0, x1), (y0, y1) = pt
            rows.append(sum((cref.int_to_limbs(P.fp_to_mont(v)) for v in (x0, x1, y0, y1)), []))
    return np.array(rows, dtype=np.uint64).reshape(-1, 16)


def g2_pts(arr):
    out = []
    for row in np.asarray(arr).reshape(-1, 16):
        v = fp_vals(row.reshape(4, 4))
        out.append(None if v == [0, 0, 0, 0] else ((v[0], v[1]), (v[2], v[3])))
    return out


def g1_from_jac(j):
    """normalised mi_g1_jac (12 limbs) -> pyref affine"""
    j = np.asarray(j).reshape(3, 4)
    x, y, z = fp_vals(j)
    if z == 0:
        return None
    assert z == 1, "MSM outputs are normalised"
    return (x, y)


def g2_from_jac(j):
    v = fp_vals(np.asarray(j).reshape(6, 4))
    if v[4] == 0 and v[5] == 0:
        return None
    assert (v[4], v[5]) == (1, 0)
    return ((v[0], v[1]), (v[2], v[3]))


def toy_pk_arrays(pk):
    """pyref.toy_setup pk dict -> numpy dict for cref.make_pk_desc / the HIP binding."""
    return {
        "log_n": pk["log_n"], "nb_public": pk["nb_public"], "nb_wires": pk["nb_wires"],
        "g1_a": g1_arr(pk["g1_a"]), "g1_b": g1_arr(pk["g1_b"]), "g1_k": g1_arr(pk["g1_k"]),
        "g1_z": g1_arr(pk["g1_z"]), "g2_b": g2_arr(pk["g2_b"]),
        "alpha1": g1_arr([pk["alpha1"]])[0], "beta1": g1_arr([pk["beta1"]])[0], "delta1": g1_arr([pk["delta1"]])[0],
        "beta2": g2_arr([pk["beta2"]])[0], "delta2": g2_arr([pk["delta2"]])[0],
        "infinity_a": np.array(pk["inf_a"], dtype=np.uint8), "infinity_b": np.array(pk["inf_b"], dtype=np.uint8),
    }


def synthetic_pk(log_n, nb_wires, nb_public, seed, inf_a_pct=10, inf_b_pct=50, n_committed=0):
    """Shape-faithful synthetic proving key: random curve points, infinity masks."""
    rng = np.random.default_rng(seed)
    inf_a = (rng.integers(0, 100, nb_wires) < inf_a_pct).astype(np.uint8)
    inf_b = (rng.integers(0, 100, nb_wires) < inf_b_pct).astype(np.uint8)
    na, nb = int((inf_a == 0).sum()), int((inf_b == 0).sum())
    committed = np.sort(rng.choice(np.arange(nb_public, nb_wires), n_committed, replace=False)).astype(np.uint32) if n_committed else None
    nk = nb_wires - nb_public - n_committed
    n = 1 << log_n
    small = cref.gen_g1(5, seed + 100)
    g2s = cref.gen_g2(2, seed + 101)
    return {
        "log_n": log_n, "nb_public": nb_public, "nb_wires": nb_wires,
        "g1_a": cref.gen_g1(na, seed + 1), "g1_b": cref.gen_g1(nb, seed + 2), "g1_k": cref.gen_g1(nk, seed + 3),
        "g1_z": cref.gen_g1(n, seed + 4), "g2_b": cref.gen_g2(nb, seed + 5),
        "alpha1": small[0], "beta1": small[1], "delta1": small[2], "beta2": g2s[0], "delta2": g2s[1],
        "infinity_a": inf_a, "infinity_b": inf_b, "committed_wires": committed,
    }
