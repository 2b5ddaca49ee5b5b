//go:build mi355x

// dumpfixture writes the fixture that turns this repository's "oracle-exact, reference-unpinned" parity into
// "pinned by the reference" (DESIGN.md section 2): one small circuit WITH a BSB22 commitment and a lookup (the two features
// of the WHIR verifier circuit that reach the prover: /root/reference/utilities/utilities.go:189, mtUtilities.go:452),
// compiled, set up, solved and proved by REAL gnark v0.11.0 on the CPU under a seeded crypto/rand, dumped as
//
//	tests/golden/gnark_pk.raw        ProvingKey.WriteRawTo                      -> mi_pk_load_raw
//	tests/golden/gnark_solution.bin  u64 LE counts, then W | a | b | c | r | s as raw fr.Element memory (4 x u64 LE Montgomery)
//	tests/golden/gnark_meta.json     nbPublic, committed wires (sorted), commitment index, hash inputs
//	tests/golden/gnark_proof.bin     proof.WriteTo                              -> expected bytes
//
// SOURCE ONLY: the build image has no Go toolchain; never compiled.  Run on first contact with Go + the pinned modules:
//
//	cd gnark-whir_amd/go/mi355x && go run -tags mi355x ./cmd/dumpfixture ../../../tests/golden
package main

import (
	"bytes"
	"crypto/rand"
	"encoding/binary"
	"encoding/json"
	"io"
	"os"
	"path/filepath"
	"sort"
	"unsafe"

	"github.com/consensys/gnark-crypto/ecc"
	"github.com/consensys/gnark-crypto/ecc/bn254/fr"
	"github.com/consensys/gnark/backend/groth16"
	groth16_bn254 "github.com/consensys/gnark/backend/groth16/bn254"
	"github.com/consensys/gnark/constraint"
	cs "github.com/consensys/gnark/constraint/bn254"
	"github.com/consensys/gnark/frontend"
	"github.com/consensys/gnark/frontend/cs/r1cs"
	"github.com/consensys/gnark/std/lookup/logderivlookup"
)

type circuit struct {
	X   frontend.Variable
	Idx frontend.Variable
	Y   frontend.Variable `gnark:",public"`
}

func (c *circuit) Define(api frontend.API) error {
	t := logderivlookup.New(api) // forces a BSB22 commitment, as utilities.go:189 does in the reference
	for i := 0; i < 8; i++ {
		t.Insert(i * i)
	}
	v := t.Lookup(c.Idx)[0]
	x3 := api.Mul(c.X, c.X, c.X)
	api.AssertIsEqual(c.Y, api.Add(x3, c.X, v))
	return nil
}

type seeded struct{ state uint64 }

func (s *seeded) Read(p []byte) (int, error) {
	for i := range p {
		s.state = s.state*6364136223846793005 + 1442695040888963407
		p[i] = byte(s.state >> 56)
	}
	return len(p), nil
}

func must(err error) {
	if err != nil {
		panic(err)
	}
}

func frBytes(v []fr.Element) []byte {
	if len(v) == 0 {
		return nil
	}
	return unsafe.Slice((*byte)(unsafe.Pointer(&v[0])), len(v)*fr.Bytes)
}

func main() {
	dir := os.Args[1]
	ccs, err := frontend.Compile(ecc.BN254.ScalarField(), r1cs.NewBuilder, &circuit{})
	must(err)
	pk, _, err := groth16.Setup(ccs)
	must(err)
	w, err := frontend.NewWitness(&circuit{X: 3, Idx: 5, Y: 3*3*3 + 3 + 25}, ecc.BN254.ScalarField())
	must(err)

	var raw bytes.Buffer
	_, err = pk.(*groth16_bn254.ProvingKey).WriteRawTo(&raw)
	must(err)
	must(os.WriteFile(filepath.Join(dir, "gnark_pk.raw"), raw.Bytes(), 0o644))

	// the proof under a seeded rand.Reader: r and s are the first two fr.SetRandom draws of prove.go
	old := rand.Reader
	rand.Reader = io.Reader(&seeded{state: 42})
	proof, err := groth16.Prove(ccs, pk, w)
	rand.Reader = old
	must(err)
	var pb bytes.Buffer
	_, err = proof.WriteTo(&pb)
	must(err)
	must(os.WriteFile(filepath.Join(dir, "gnark_proof.bin"), pb.Bytes(), 0o644))

	// the same draws again for the fixture's (r, s)
	rand.Reader = io.Reader(&seeded{state: 42})
	var r, s fr.Element
	_, err = r.SetRandom()
	must(err)
	_, err = s.SetRandom()
	must(err)
	rand.Reader = old

	// solution vectors: gnark's solver, the commitment hint computed by gnark itself
	sol, err := ccs.(*cs.R1CS).Solve(w)
	must(err)
	so := sol.(*cs.R1CSSolution)
	var sb bytes.Buffer
	for _, n := range []int{len(so.W), len(so.A)} {
		must(binary.Write(&sb, binary.LittleEndian, uint64(n)))
	}
	for _, v := range [][]fr.Element{so.W, so.A, so.B, so.C, {r}, {s}} {
		sb.Write(frBytes(v))
	}
	must(os.WriteFile(filepath.Join(dir, "gnark_solution.bin"), sb.Bytes(), 0o644))

	info := ccs.(*cs.R1CS).CommitmentInfo.(constraint.Groth16Commitments)
	removed := append([]int{}, info.CommitmentIndexes()...)
	for _, p := range info.GetPrivateCommitted() {
		removed = append(removed, p...)
	}
	sort.Ints(removed)
	meta := map[string]any{"nb_public": ccs.GetNbPublicVariables(), "removed_from_k": removed, "nb_commitments": len(info),
		"gnark": "v0.11.0", "gnark_crypto": "v0.14.1-0.20241217131346-b998989abdbe"}
	mb, _ := json.MarshalIndent(meta, "", " ")
	must(os.WriteFile(filepath.Join(dir, "gnark_meta.json"), mb, 0o644))
}
