//go:build mi355x

package mi355x

// Parity harness of INTEGRATION.md section 4 -- SOURCE ONLY (no Go toolchain in the build image; never compiled).
// On a machine with Go, the pinned modules (gnark v0.11.0, gnark-crypto v0.14.1-0.20241217131346-b998989abdbe) and an
// MI355X:   go test -tags mi355x ./...
// It proves one small circuit with gnark's CPU prover and with this package on the same (ccs, pk, witness) under the same
// deterministic randomness and requires byte-identical proof.WriteTo output -- the check that would turn "parity
// unpinned" (DESIGN.md section 2) into "pinned by the reference".

import (
	"bytes"
	"crypto/rand"
	"io"
	"testing"

	"github.com/consensys/gnark-crypto/ecc"
	"github.com/consensys/gnark/backend/groth16"
	groth16_bn254 "github.com/consensys/gnark/backend/groth16/bn254"
	cs "github.com/consensys/gnark/constraint/bn254"
	"github.com/consensys/gnark/frontend"
	"github.com/consensys/gnark/frontend/cs/r1cs"
	"github.com/consensys/gnark/std/lookup/logderivlookup"
)

// cubic: x^3 + x + 5 == y (no commitment); committed (below) adds api.Commit over a public and a private wire plus a lookup,
// i.e. the Pedersen / SerializeCommitment / BatchProve path of prove.go that the WHIR verifier circuit takes.
type cubic struct {
	X frontend.Variable
	Y frontend.Variable `gnark:",public"`
}

func (c *cubic) Define(api frontend.API) error {
	x3 := api.Mul(c.X, c.X, c.X)
	api.AssertIsEqual(c.Y, api.Add(x3, c.X, 5))
	return nil
}

type committed struct {
	X   frontend.Variable
	Idx frontend.Variable
	Y   frontend.Variable `gnark:",public"`
}

func (c *committed) Define(api frontend.API) error {
	t := logderivlookup.New(api) // what /root/reference/utilities/utilities.go:189 does: forces a BSB22 commitment
	for i := 0; i < 8; i++ {
		t.Insert(i * i)
	}
	v := t.Lookup(c.Idx)[0]
	cm, err := api.(frontend.Committer).Commit(c.X, c.Y) // one private and one PUBLIC committed wire: the hashed prefix is not empty
	if err != nil {
		return err
	}
	api.AssertIsDifferent(cm, 0)
	api.AssertIsEqual(c.Y, api.Add(api.Mul(c.X, c.X, c.X), c.X, v))
	return nil
}

// seeded stream standing in for crypto/rand.Reader so both provers draw the same (r, s)
type seeded struct{ state uint64 }

func (s *seeded) Read(p []byte) (int, error) {
	for i := range p {
		s.state = s.state*6364136223846793005 + 1442695040888963407
		p[i] = byte(s.state >> 56)
	}
	return len(p), nil
}

func withSeed(seed uint64, f func()) {
	old := rand.Reader
	rand.Reader = io.Reader(&seeded{state: seed})
	defer func() { rand.Reader = old }()
	f()
}

func TestProofBytesMatchGnarkCPU(t *testing.T) {
	t.Run("cubic", func(t *testing.T) { proveBoth(t, &cubic{}, &cubic{X: 3, Y: 35}) })
	t.Run("commit+lookup", func(t *testing.T) { proveBoth(t, &committed{}, &committed{X: 3, Idx: 5, Y: 27 + 3 + 25}) })
}

func proveBoth(t *testing.T, circuit, assignment frontend.Circuit) {
	ccs, err := frontend.Compile(ecc.BN254.ScalarField(), r1cs.NewBuilder, circuit)
	if err != nil {
		t.Fatal(err)
	}
	pk, _, err := groth16.Setup(ccs)
	if err != nil {
		t.Fatal(err)
	}
	w, err := frontend.NewWitness(assignment, ecc.BN254.ScalarField())
	if err != nil {
		t.Fatal(err)
	}
	var cpu, gpu bytes.Buffer
	withSeed(42, func() {
		p, err := groth16.Prove(ccs, pk, w)
		if err != nil {
			t.Fatal(err)
		}
		if _, err := p.WriteTo(&cpu); err != nil {
			t.Fatal(err)
		}
	})
	withSeed(42, func() {
		gpk := &ProvingKey{ProvingKey: *pk.(*groth16_bn254.ProvingKey)}
		p, err := Prove(ccs.(*cs.R1CS), gpk, w)
		if err != nil {
			t.Fatal(err)
		}
		if _, err := p.WriteTo(&gpu); err != nil {
			t.Fatal(err)
		}
	})
	if !bytes.Equal(cpu.Bytes(), gpu.Bytes()) {
		t.Fatalf("proof bytes differ:\ncpu %x\ngpu %x", cpu.Bytes(), gpu.Bytes())
	}
}
