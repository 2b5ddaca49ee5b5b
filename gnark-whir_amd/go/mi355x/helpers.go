//go:build mi355x

package mi355x

// Helpers of the shim (SOURCE ONLY, never compiled here — see mi355x.go).

import (
	"sort"

	"github.com/consensys/gnark-crypto/ecc/bn254"
	"github.com/consensys/gnark-crypto/ecc/bn254/fr"
	"github.com/consensys/gnark-crypto/ecc/bn254/fr/pedersen"
)

func flatten(xs [][]int) []int {
	var out []int
	for _, x := range xs {
		out = append(out, x...)
	}
	return out
}

func sortedUint32(xs []int) []uint32 {
	sort.Ints(xs)
	out := make([]uint32, 0, len(xs))
	for i, x := range xs {
		if i == 0 || x != xs[i-1] {
			out = append(out, uint32(x))
		}
	}
	return out
}

// foldedPok reproduces prove.go's CommitmentPok: one ProveKnowledge per commitment key, folded with
// the challenge fr.Hash(commitments, "G16-BSB22").  O(#committed wires); kept on gnark's CPU code.
func foldedPok(pk *ProvingKey, vals [][]fr.Element, commitments []bn254.G1Affine) (bn254.G1Affine, error) {
	poks := make([]bn254.G1Affine, len(pk.CommitmentKeys))
	for i := range pk.CommitmentKeys {
		p, err := pk.CommitmentKeys[i].ProveKnowledge(vals[i])
		if err != nil {
			return bn254.G1Affine{}, err
		}
		poks[i] = p
	}
	buf := make([]byte, 0, len(commitments)*fr.Bytes*2)
	for i := range commitments {
		b := commitments[i].Marshal()
		buf = append(buf, b...)
	}
	challenge, err := fr.Hash(buf, []byte("G16-BSB22"), 1)
	if err != nil {
		return bn254.G1Affine{}, err
	}
	return pedersen.Fold(poks, challenge[0], nil...)
}
