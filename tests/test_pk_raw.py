"""proving-key ingestion from gnark's ProvingKey.WriteRawTo stream (layout recalled, unverified -- see oracle/pk_raw.py
and csrc/pk_raw.hip).  CPU part: the host-side walk over the stream (mi_pk_raw_inspect) on streams written by the oracle-side
writer.  GPU part: mi_pk_load_raw (device conversion big-endian canonical -> Montgomery limbs) and a proof from the loaded key,
byte for byte against the oracle's proof from the original key."""
import numpy as np
import pytest
import pyref as P
import cref
import pk_raw
from helpers import *
from gpu_common import load_binding


def _toy(nc=200, npub=5, seed=31, n_ped=0):
    cs = P.ToyR1CS(nc, npub, seed); td = P.ToyTrapdoor(seed)
    pk, exps, dom = P.toy_setup(cs, td)
    keys = []
    for k in range(n_ped):
        basis = g1_pts(cref.gen_g1(7 + k, 40 + k)); bes = g1_pts(cref.gen_g1(7 + k, 50 + k))
        keys.append((basis, bes))
    return cs, pk, dom, keys


def test_raw_stream_inspect_counts_and_offsets():
    B = load_binding()
    cs, pk, dom, keys = _toy(n_ped=2)
    blob = pk_raw.write_pk_raw(pk, keys)
    info = B.pk_raw_inspect(blob)
    assert info.log_n == pk["log_n"] and info.nb_wires == cs.nb_wires and info.n_commitment_keys == 2
    assert (info.n_g1_a, info.n_g1_b, info.n_g1_k, info.n_g1_z, info.n_g2_b) == tuple(len(pk[k]) for k in ("g1_a", "g1_b", "g1_k", "g1_z", "g2_b"))
    assert info.off_alpha1 == 8 + 5 * 32 + 1 and info.off_g1_a == info.off_alpha1 + 3 * 64 + 4
    assert blob[info.off_g1_a:info.off_g1_a + 64] == pk_raw.g1_raw(pk["g1_a"][0])
    assert blob[info.off_g2_b:info.off_g2_b + 128] == pk_raw.g2_raw(pk["g2_b"][0])
    assert list(info.n_basis[:2]) == [7, 8]
    # strictness: truncated, padded, or internally inconsistent streams are refused
    for bad in (blob[:-1], blob + b"\x00", blob[:100], b""):
        with pytest.raises(B.MiError):
            B.pk_raw_inspect(bad)
    broken = bytearray(blob); broken[info.off_g1_a - 1] ^= 1        # len(G1.A) no longer matches InfinityA
    with pytest.raises(B.MiError):
        B.pk_raw_inspect(bytes(broken))
    broken = bytearray(blob); broken[7] = 3                         # cardinality not a power of two
    with pytest.raises(B.MiError):
        B.pk_raw_inspect(bytes(broken))


@pytest.mark.gpu
@pytest.mark.parametrize("nc,npub,seed", [(200, 5, 31), (1000, 3, 77), (33, 33, 5)])
def test_load_raw_and_prove_vs_oracle(nc, npub, seed):
    B = load_binding()
    ctx = B.Context(0)
    try:
        cs, pk, dom, keys = _toy(nc, npub, seed, n_ped=1)
        blob = pk_raw.write_pk_raw(pk, keys)
        pkh, peds = ctx.pk_load_raw(blob, cs.nb_public)
        w, a, b, c = cs.solve()
        r, s = 123456789 + seed, 987654321 + seed
        W, A_, B_, C_ = fr_arr(w), fr_arr(a), fr_arr(b), fr_arr(c)
        got, _ = ctx.prove(pkh, W, A_, B_, C_, fr_arr([r])[0], fr_arr([s])[0])
        want = P.toy_prove(cs, pk, dom, r, s)
        assert B.proof_write(got["raw"]) == P.proof_bytes(want)
        # the same key through the array entry point gives the same proof
        pkh2 = ctx.pk_load(toy_pk_arrays(pk))
        again, _ = ctx.prove(pkh2, W, A_, B_, C_, fr_arr([r])[0], fr_arr([s])[0])
        assert np.array_equal(again["raw"], got["raw"])
        # the Pedersen key that came with the stream
        basis, bes = keys[0]
        vals = cref.gen_scalars(len(basis), 9, 1)
        assert len(peds) == 1
        assert np.array_equal(ctx.pedersen_commit(peds[0], vals), cref.pedersen_msm(g1_arr(basis), vals))
        assert np.array_equal(ctx.pedersen_commit(peds[0], vals, knowledge=True), cref.pedersen_msm(g1_arr(bes), vals))
        ctx.pedersen_pk_free(peds[0]); ctx.pk_free(pkh); ctx.pk_free(pkh2)
        # a compressed-flag byte inside a point section is refused on the device side
        info = B.pk_raw_inspect(blob)
        broken = bytearray(blob); broken[info.off_g1_z] |= 0x80
        with pytest.raises(B.MiError):
            ctx.pk_load_raw(bytes(broken), cs.nb_public)
    finally:
        ctx.close()
