"""Writer of gnark v0.11.0's groth16 `ProvingKey.WriteRawTo` stream for the BN254 backend -- TEST INFRASTRUCTURE
(oracle/, like pyref.py): it feeds `mi_pk_load_raw` (csrc/pk_raw.hip) in the tests.

LAYOUT RECALLED, UNVERIFIED.  The reference never writes a key (it re-runs groth16.Setup on every run, /root/reference/mt.go:448)
and holds no key file; gnark and gnark-crypto are absent from this image (no Go toolchain, go.mod:6-7 pins them by version
only).  The byte layout below is restated from the published behaviour of gnark backend/groth16/bn254/marshal.go
(`(*ProvingKey).writeTo(w, raw=true)`) and gnark-crypto ecc/bn254/marshal.go (`Encoder` with `RawEncoding()`), fr/fft
`Domain.WriteTo`, fr/pedersen `ProvingKey.WriteRawTo`.  First contact with real gnark must check it with
gnark-whir_amd/go/mi355x/cmd/dumpfixture (writes a real key + proof into tests/golden/).

    Domain        u64 BE Cardinality | fr CardinalityInv | fr Generator | fr GeneratorInv | fr FrMultiplicativeGen |
                  fr FrMultiplicativeGenInv | u8 withPrecompute              (fr = 32 bytes big-endian, canonical)
    G1.Alpha, G1.Beta, G1.Delta                       3 x 64 B   raw G1: X | Y big-endian canonical, flags in the two top
                                                                 bits of byte 0: 00 = uncompressed, 01 = infinity
    G1.A, G1.B, G1.Z, G1.K                            each: u32 BE count | count x 64 B
    G2.Beta, G2.Delta                                 2 x 128 B  raw G2: X.A1 | X.A0 | Y.A1 | Y.A0
    G2.B                                              u32 BE count | count x 128 B
    nbWires u64 BE | NbInfinityA u64 BE | NbInfinityB u64 BE
    InfinityA, InfinityB                              each: u32 BE count | ceil(count / 8) bytes, element i = bit (7 - i % 8) of byte i / 8
    u32 BE number of commitment keys, then per key (pedersen.ProvingKey.WriteRawTo):
                                                      Basis: u32 BE count | count x 64 B ; BasisExpSigma: u32 BE count | count x 64 B
"""
import struct
import pyref as P


def _fr(x):
    return int(x % P.R_MOD).to_bytes(32, "big")


def g1_raw(pt):
    if pt is None:
        return bytes([0x40]) + bytes(63)
    return int(pt[0]).to_bytes(32, "big") + int(pt[1]).to_bytes(32, "big")


def g2_raw(pt):
    if pt is None:
        return bytes([0x40]) + bytes(127)
    (x0, x1), (y0, y1) = pt
    return b"".join(int(v).to_bytes(32, "big") for v in (x1, x0, y1, y0))


def _g1s(pts):
    return struct.pack(">I", len(pts)) + b"".join(g1_raw(p) for p in pts)


def _g2s(pts):
    return struct.pack(">I", len(pts)) + b"".join(g2_raw(p) for p in pts)


def _bools(bs):
    out = bytearray((len(bs) + 7) // 8)
    for i, b in enumerate(bs):
        if b:
            out[i // 8] |= 1 << (7 - i % 8)
    return struct.pack(">I", len(bs)) + bytes(out)


def write_pk_raw(pk, commitment_keys=()):
    """pk: pyref.toy_setup's dict (affine points as integer tuples).  commitment_keys: [(basis, basis_exp_sigma), ...]"""
    n = 1 << pk["log_n"]
    w = pow(P.FR_ROOT_2_28, 1 << (28 - pk["log_n"]), P.R_MOD)
    out = [struct.pack(">Q", n), _fr(pow(n, -1, P.R_MOD)), _fr(w), _fr(pow(w, -1, P.R_MOD)), _fr(5), _fr(pow(5, -1, P.R_MOD)), b"\x01"]
    out += [g1_raw(pk["alpha1"]), g1_raw(pk["beta1"]), g1_raw(pk["delta1"])]
    out += [_g1s(pk["g1_a"]), _g1s(pk["g1_b"]), _g1s(pk["g1_z"]), _g1s(pk["g1_k"])]
    out += [g2_raw(pk["beta2"]), g2_raw(pk["delta2"]), _g2s(pk["g2_b"])]
    out += [struct.pack(">QQQ", pk["nb_wires"], sum(map(bool, pk["inf_a"])), sum(map(bool, pk["inf_b"])))]
    out += [_bools(pk["inf_a"]), _bools(pk["inf_b"]), struct.pack(">I", len(commitment_keys))]
    for basis, bes in commitment_keys:
        out += [_g1s(basis), _g1s(bes)]
    return b"".join(out)
