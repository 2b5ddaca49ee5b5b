//go:build mi355x

// Package mi355x is the cgo shim that lets gnark's groth16.Prove (the call at
// reilabs/gnark-whir mt.go:496) run its post-solve work on an AMD MI355X through
// libmi355x_groth16.so (include/mi355x_groth16.h).
//
// STATUS: SOURCE ONLY.  The build container has no Go toolchain and no module cache
// (gnark v0.11.0, gnark-crypto v0.14.1-0.20241217131346-b998989abdbe, go.mod:6-7), so this file
// has never been compiled.  It is written against the public API of those pinned versions and
// shaped like gnark's own accelerator package backend/groth16/bn254/icicle (device-resident
// ProvingKey + Prove).  Everything numeric happens behind the C-ABI, which IS tested.
package mi355x

/*
#cgo CFLAGS: -I${SRCDIR}/../../../include
#cgo LDFLAGS: -L${SRCDIR}/../.. -lmi355x_groth16 -Wl,-rpath,${SRCDIR}/../..
#include <stdlib.h>
#include "mi355x_groth16.h"
*/
import "C"

import (
	"errors"
	"fmt"
	"math/big"
	"runtime"
	"sync"
	"unsafe"

	"github.com/consensys/gnark-crypto/ecc/bn254"
	"github.com/consensys/gnark-crypto/ecc/bn254/fr"
	"github.com/consensys/gnark-crypto/ecc/bn254/fr/hash_to_field"
	"github.com/consensys/gnark/backend"
	groth16_bn254 "github.com/consensys/gnark/backend/groth16/bn254"
	"github.com/consensys/gnark/backend/witness"
	"github.com/consensys/gnark/constraint"
	cs "github.com/consensys/gnark/constraint/bn254"
	"github.com/consensys/gnark/constraint/solver"
)

// ProvingKey embeds gnark's key and the device-resident handle (cf. icicle_bn254.ProvingKey).
type ProvingKey struct {
	groth16_bn254.ProvingKey
	once sync.Once
	ctx  *C.mi_ctx    // key load + the mid-solve Pedersen commitments (guarded by mu: a context is single-threaded)
	mu   sync.Mutex
	pool *C.mi_prover // InFlight contexts on the same GPU: concurrent Prove calls overlap on the device
	dev  *C.mi_pk     // device-resident key, read-only during prove, shared by every context
	ped  []*C.mi_pedersen_pk
	err  error
}

// InFlight is the number of proofs the device keeps in flight per key (mi_prover_create).  Goroutines calling Prove
// beyond that queue up in the library.  3 fills an MI355X at the WHIR-verifier size (DESIGN.md section 5).
var InFlight = 3

func status(ctx *C.mi_ctx, rc C.int32_t) error {
	if rc == C.MI_OK {
		return nil
	}
	return fmt.Errorf("mi355x: rc=%d: %s", int(rc), C.GoString(C.mi_last_error(ctx)))
}

// setup uploads the key once (mi_pk_load); later proofs reuse it.
func (pk *ProvingKey) setup(r1cs *cs.R1CS, device int) error {
	pk.once.Do(func() {
		if rc := C.mi_init(C.int(device), &pk.ctx); rc != C.MI_OK {
			pk.err = fmt.Errorf("mi355x: mi_init rc=%d (no gfx950 device?)", int(rc))
			return
		}
		if rc := C.mi_prover_create(C.int(device), C.uint32_t(InFlight), &pk.pool); rc != C.MI_OK {
			pk.err = fmt.Errorf("mi355x: mi_prover_create rc=%d", int(rc))
			return
		}
		var d C.mi_pk_desc
		d.log_n = C.uint32_t(log2(pk.Domain.Cardinality))
		d.nb_public = C.uint32_t(r1cs.GetNbPublicVariables())
		d.nb_wires = C.uint64_t(len(pk.InfinityA))
		d.g1_a, d.n_g1_a = (*C.mi_g1_affine)(unsafe.Pointer(&pk.G1.A[0])), C.uint64_t(len(pk.G1.A))
		d.g1_b, d.n_g1_b = (*C.mi_g1_affine)(unsafe.Pointer(&pk.G1.B[0])), C.uint64_t(len(pk.G1.B))
		d.g1_k, d.n_g1_k = (*C.mi_g1_affine)(unsafe.Pointer(&pk.G1.K[0])), C.uint64_t(len(pk.G1.K))
		d.g1_z, d.n_g1_z = (*C.mi_g1_affine)(unsafe.Pointer(&pk.G1.Z[0])), C.uint64_t(len(pk.G1.Z))
		d.g2_b, d.n_g2_b = (*C.mi_g2_affine)(unsafe.Pointer(&pk.G2.B[0])), C.uint64_t(len(pk.G2.B))
		d.alpha1 = *(*C.mi_g1_affine)(unsafe.Pointer(&pk.G1.Alpha))
		d.beta1 = *(*C.mi_g1_affine)(unsafe.Pointer(&pk.G1.Beta))
		d.delta1 = *(*C.mi_g1_affine)(unsafe.Pointer(&pk.G1.Delta))
		d.beta2 = *(*C.mi_g2_affine)(unsafe.Pointer(&pk.G2.Beta))
		d.delta2 = *(*C.mi_g2_affine)(unsafe.Pointer(&pk.G2.Delta))
		// []bool is one byte per element in Go's memory model
		d.infinity_a = (*C.uint8_t)(unsafe.Pointer(&pk.InfinityA[0]))
		d.infinity_b = (*C.uint8_t)(unsafe.Pointer(&pk.InfinityB[0]))
		// wires removed from the K MSM: private committed + commitment wires (prove.go "toRemove")
		info := r1cs.CommitmentInfo.(constraint.Groth16Commitments)
		removed := sortedUint32(append(flatten(info.GetPrivateCommitted()), info.CommitmentIndexes()...))
		if len(removed) > 0 {
			d.committed_wires, d.n_committed = (*C.uint32_t)(unsafe.Pointer(&removed[0])), C.uint64_t(len(removed))
		}
		pk.err = status(pk.ctx, C.mi_pk_load(pk.ctx, &d, &pk.dev))
		runtime.KeepAlive(removed)
	})
	return pk.err
}

// Prove mirrors groth16_bn254.Prove: solve on the CPU (gnark's solver, incl. the reference's
// utilities.IndexOf hint passed through opts), everything after it on the GPU.
func Prove(r1cs *cs.R1CS, pk *ProvingKey, fullWitness witness.Witness, opts ...backend.ProverOption) (*groth16_bn254.Proof, error) {
	opt, err := backend.NewProverConfig(opts...)
	if err != nil {
		return nil, err
	}
	if err := pk.setup(r1cs, 0); err != nil {
		return nil, err
	}
	commitmentInfo := r1cs.CommitmentInfo.(constraint.Groth16Commitments)
	proof := &groth16_bn254.Proof{Commitments: make([]bn254.G1Affine, len(commitmentInfo))}
	privateCommittedValues := make([][]fr.Element, len(commitmentInfo))

	// BSB22 commitment hint: the Pedersen MSM runs mid-solve through the same device MSM (SURVEY 8f N1)
	solverOpts := opt.SolverOpts[:len(opt.SolverOpts):len(opt.SolverOpts)]
	solverOpts = append(solverOpts, solver.OverrideHint(commitmentInfo.GetHintID(0), func(_ *big.Int, in []*big.Int, out []*big.Int) error {
		// NOTE: one override per commitment in gnark; shown for the single-commitment WHIR circuit
		i := 0
		nPriv := len(commitmentInfo[i].PrivateCommitted)
		vals := make([]fr.Element, nPriv)
		for j, v := range in[len(in)-nPriv:] {
			vals[j].SetBigInt(v)
		}
		privateCommittedValues[i] = vals
		// device-resident Pedersen key (mi_pedersen_pk_load once per key, see pedersenKey below)
		pk.mu.Lock()
		defer pk.mu.Unlock()
		ppk, err := pk.pedersenKey(i)
		if err != nil {
			return err
		}
		if rc := C.mi_pedersen_commit(pk.ctx, ppk, (*C.mi_fr)(unsafe.Pointer(&vals[0])), C.size_t(len(vals)),
			(*C.mi_g1_affine)(unsafe.Pointer(&proof.Commitments[i]))); rc != C.MI_OK {
			return status(pk.ctx, rc)
		}
		// challenge = HashToField(commitment || public committed), DST "bsb22-commitment" (stays in Go)
		hashed, err := fr.Hash(proof.Commitments[i].Marshal(), []byte(constraint.CommitmentDst), 1)
		if err != nil {
			return err
		}
		hashed[0].BigInt(out[0])
		return nil
	}))

	_solution, err := r1cs.Solve(fullWitness, solverOpts...)
	if err != nil {
		return nil, err
	}
	solution := _solution.(*cs.R1CSSolution)
	W, a, b, c := []fr.Element(solution.W), []fr.Element(solution.A), []fr.Element(solution.B), []fr.Element(solution.C)

	// CommitmentPok (ProveKnowledge + Fold) stays on gnark's CPU code path: O(#committed) work.
	if len(commitmentInfo) > 0 {
		if proof.CommitmentPok, err = foldedPok(pk, privateCommittedValues, proof.Commitments); err != nil {
			return nil, err
		}
	}

	// same sampling order as prove.go so a test that swaps rand.Reader gets byte-identical proofs
	var r, s fr.Element
	if _, err := r.SetRandom(); err != nil {
		return nil, err
	}
	if _, err := s.SetRandom(); err != nil {
		return nil, err
	}

	// submit + wait on the pool: this goroutine blocks in cgo (the Go scheduler parks it on its own OS thread) while the
	// proofs of other goroutines overlap with it on the GPU.  The output lives in C memory: the library writes it from a
	// worker thread after this cgo call has returned, which Go memory passed by pointer must not be used for.
	out := (*C.mi_proof_out)(C.malloc(C.size_t(unsafe.Sizeof(C.mi_proof_out{}))))
	defer C.free(unsafe.Pointer(out))
	var pin runtime.Pinner // W, a, b, c are read by the worker thread until mi_prover_wait returns
	pin.Pin(&W[0]); pin.Pin(&a[0]); pin.Pin(&b[0]); pin.Pin(&c[0])
	defer pin.Unpin()
	var ticket C.uint64_t
	rc := C.mi_prover_submit(pk.pool, pk.dev,
		(*C.mi_fr)(unsafe.Pointer(&W[0])), C.size_t(len(W)),
		(*C.mi_fr)(unsafe.Pointer(&a[0])), (*C.mi_fr)(unsafe.Pointer(&b[0])), (*C.mi_fr)(unsafe.Pointer(&c[0])), C.size_t(len(a)),
		(*C.mi_fr)(unsafe.Pointer(&r)), (*C.mi_fr)(unsafe.Pointer(&s)), out, nil, &ticket)
	if rc != C.MI_OK {
		return nil, fmt.Errorf("mi355x: mi_prover_submit rc=%d", int(rc))
	}
	if rc := C.mi_prover_wait(pk.pool, ticket); rc != C.MI_OK {
		return nil, fmt.Errorf("mi355x: rc=%d: %s", int(rc), C.GoString(C.mi_prover_last_error(pk.pool)))
	}
	runtime.KeepAlive(W)
	proof.Ar = *(*bn254.G1Affine)(unsafe.Pointer(&out.ar))
	proof.Bs = *(*bn254.G2Affine)(unsafe.Pointer(&out.bs))
	proof.Krs = *(*bn254.G1Affine)(unsafe.Pointer(&out.krs))
	return proof, nil
}

// pedersenKey uploads CommitmentKeys[i].Basis / BasisExpSigma once (mi_pedersen_pk_load).
func (pk *ProvingKey) pedersenKey(i int) (*C.mi_pedersen_pk, error) {
	if pk.ped == nil {
		pk.ped = make([]*C.mi_pedersen_pk, len(pk.CommitmentKeys))
	}
	if pk.ped[i] == nil {
		k := &pk.CommitmentKeys[i]
		rc := C.mi_pedersen_pk_load(pk.ctx, (*C.mi_g1_affine)(unsafe.Pointer(&k.Basis[0])),
			(*C.mi_g1_affine)(unsafe.Pointer(&k.BasisExpSigma[0])), C.size_t(len(k.Basis)), &pk.ped[i])
		if err := status(pk.ctx, rc); err != nil {
			return nil, err
		}
	}
	return pk.ped[i], nil
}

func log2(n uint64) int { k := 0; for (uint64(1) << k) < n { k++ }; return k }

var _ = errors.New
var _ = hash_to_field.New
