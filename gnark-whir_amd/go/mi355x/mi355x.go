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
	fcs "github.com/consensys/gnark/frontend/cs"
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
		pk.err = pk.withKeyDesc(r1cs, func(d *C.mi_pk_desc) error {
			return status(pk.ctx, C.mi_pk_load(pk.ctx, d, &pk.dev))
		})
	})
	return pk.err
}

// withKeyDesc builds the mi_pk_desc of this key and hands it to fn (mi_pk_load, mi_pk_load_sharded).
//
// cgo pointer passing: the descriptor holds POINTERS to the key's Go slices, so it must not be a Go variable handed to C by
// address -- cgocheck rejects "Go pointer to unpinned Go pointer".  The descriptor therefore lives in C memory (C.calloc) and every
// Go array it points to is pinned with a runtime.Pinner for the duration of fn (Go >= 1.21: a Go pointer may be stored in C memory
// while its target is pinned).  The library reads the arrays during the call only and keeps none of these pointers.
func (pk *ProvingKey) withKeyDesc(r1cs *cs.R1CS, fn func(d *C.mi_pk_desc) error) error {
	if len(pk.InfinityA) == 0 || len(pk.InfinityB) != len(pk.InfinityA) {
		return errors.New("mi355x: proving key without wires")
	}
	d := (*C.mi_pk_desc)(C.calloc(1, C.size_t(unsafe.Sizeof(C.mi_pk_desc{}))))
	if d == nil {
		return errors.New("mi355x: out of memory")
	}
	defer C.free(unsafe.Pointer(d))
	var pin runtime.Pinner
	defer pin.Unpin()
	d.log_n = C.uint32_t(log2(pk.Domain.Cardinality))
	d.nb_public = C.uint32_t(r1cs.GetNbPublicVariables())
	d.nb_wires = C.uint64_t(len(pk.InfinityA))
	// &s[0] panics on an empty slice (all-public circuits have no K points, tiny ones may lack A or B): g1 / g2 guard it and pin
	d.g1_a, d.n_g1_a = g1(&pin, pk.G1.A), C.uint64_t(len(pk.G1.A))
	d.g1_b, d.n_g1_b = g1(&pin, pk.G1.B), C.uint64_t(len(pk.G1.B))
	d.g1_k, d.n_g1_k = g1(&pin, pk.G1.K), C.uint64_t(len(pk.G1.K))
	d.g1_z, d.n_g1_z = g1(&pin, pk.G1.Z), C.uint64_t(len(pk.G1.Z))
	d.g2_b, d.n_g2_b = g2(&pin, pk.G2.B), C.uint64_t(len(pk.G2.B))
	d.alpha1 = *(*C.mi_g1_affine)(unsafe.Pointer(&pk.G1.Alpha)) // copied by value: no pointer kept
	d.beta1 = *(*C.mi_g1_affine)(unsafe.Pointer(&pk.G1.Beta))
	d.delta1 = *(*C.mi_g1_affine)(unsafe.Pointer(&pk.G1.Delta))
	d.beta2 = *(*C.mi_g2_affine)(unsafe.Pointer(&pk.G2.Beta))
	d.delta2 = *(*C.mi_g2_affine)(unsafe.Pointer(&pk.G2.Delta))
	// []bool is one byte per element in Go's memory model
	pin.Pin(&pk.InfinityA[0])
	pin.Pin(&pk.InfinityB[0])
	d.infinity_a = (*C.uint8_t)(unsafe.Pointer(&pk.InfinityA[0]))
	d.infinity_b = (*C.uint8_t)(unsafe.Pointer(&pk.InfinityB[0]))
	// wires removed from the K MSM: private committed + commitment wires (prove.go "toRemove")
	info := r1cs.CommitmentInfo.(constraint.Groth16Commitments)
	removed := sortedUint32(append(flatten(info.GetPrivateCommitted()), info.CommitmentIndexes()...))
	if len(removed) > 0 {
		pin.Pin(&removed[0])
		d.committed_wires, d.n_committed = (*C.uint32_t)(unsafe.Pointer(&removed[0])), C.uint64_t(len(removed))
	}
	return fn(d)
}

// Every C call of this package and the rule that makes its pointer arguments legal (cmd/cgo "Passing pointers"):
//   mi_init(&pk.ctx), mi_prover_create(&pk.pool), mi_pk_load(.., &pk.dev), mi_pedersen_pk_load(.., &pk.ped[i]), mi_group_create(&g),
//   mi_pk_load_sharded(.., &gr.spk)      pointer to ONE field / element that receives a C pointer: the memory passed is that field
//                                        (or a slice backing array of C pointers) and holds no Go pointer
//   mi_pk_load / mi_pk_load_sharded(d)   d is C memory; the Go arrays it points to are pinned (withKeyDesc)
//   mi_prover_submit(W, a, b, c, &r, &s, out, &ticket)
//                                        W, a, b, c: flat []fr.Element backing arrays (no pointers inside), pinned because a worker
//                                        thread reads them after the call returns; r, s, ticket: Go values without pointers, used
//                                        during the call only; out: C memory (written after the call returns)
//   mi_groth16_prove_sharded(W, a, b, c, r, s, &out)
//                                        flat arrays / values without pointers, used during the call only
//   mi_pedersen_commit / _prove_knowledge / _fold
//                                        flat []fr.Element / []bn254.G1Affine backing arrays and values, during the call only
//   mi_group_create(&ids[0])             []C.int backing array
//   mi_last_error / mi_prover_last_error / mi_group_last_error
//                                        C strings owned by the library, copied at once with C.GoString
//
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

	// BSB22 commitment hint, as gnark v0.11.0 prove.go registers it: ONE override on the placeholder hint id; in[0] is the
	// commitment's index i, the next len(PublicAndCommitmentCommitted) inputs are hashed with the commitment, the rest are the
	// private committed values.  (gnark <= v0.9 registered one override per commitmentInfo[i].HintID instead: if this does
	// not compile against the module cache at hand, that is the other shape -- same body, i captured per closure.)
	// The Pedersen MSM runs mid-solve on the device (mi_pedersen_commit, SURVEY 8f N1); the hash-to-field stays in Go and uses
	// the CONFIGURED opt.HashToFieldFn over constraint.SerializeCommitment(commitment, hashed, ...), exactly like prove.go --
	// round 1's shim hashed only the commitment point, which breaks every circuit with public committed inputs (the WHIR
	// circuit has them: mtUtilities.go:92,452).
	solverOpts := opt.SolverOpts[:len(opt.SolverOpts):len(opt.SolverOpts)]
	if len(commitmentInfo) > 0 {
		bsb22ID := solver.GetHintID(fcs.Bsb22CommitmentComputePlaceholder)
		solverOpts = append(solverOpts, solver.OverrideHint(bsb22ID, func(_ *big.Int, in []*big.Int, out []*big.Int) error {
			i := int(in[0].Int64())
			if i < 0 || i >= len(commitmentInfo) {
				return fmt.Errorf("mi355x: commitment index %d out of range", i)
			}
			in = in[1:]
			hashed := in[:len(commitmentInfo[i].PublicAndCommitmentCommitted)]
			committed := in[len(hashed):]
			vals := make([]fr.Element, len(commitmentInfo[i].PrivateCommitted))
			for j, inJ := range committed {
				vals[j].SetBigInt(inJ)
			}
			privateCommittedValues[i] = vals
			if len(vals) > 0 { // an empty commitment is the point at infinity (the zero value already there)
				pk.mu.Lock()
				ppk, err := pk.pedersenKey(i)
				pk.mu.Unlock()
				if err != nil {
					return err
				}
				// mi_prover_commit: synchronous, safe from any goroutine (the pool serialises the commitments of all callers
				// on a context of its own, ahead of the proofs in flight)
				if rc := C.mi_prover_commit(pk.pool, ppk, (*C.mi_fr)(unsafe.Pointer(&vals[0])), C.size_t(len(vals)),
					(*C.mi_g1_affine)(unsafe.Pointer(&proof.Commitments[i]))); rc != C.MI_OK {
					return fmt.Errorf("mi355x: mi_prover_commit rc=%d: %s", int(rc), C.GoString(C.mi_prover_last_error(pk.pool)))
				}
			}
			opt.HashToFieldFn.Write(constraint.SerializeCommitment(proof.Commitments[i].Marshal(), hashed, (fr.Bits-1)/8+1))
			hashBts := opt.HashToFieldFn.Sum(nil)
			opt.HashToFieldFn.Reset()
			nbBuf := fr.Bytes
			if opt.HashToFieldFn.Size() < fr.Bytes {
				nbBuf = opt.HashToFieldFn.Size()
			}
			var res fr.Element
			res.SetBytes(hashBts[:nbBuf])
			res.BigInt(out[0])
			return nil
		}))
	}

	_solution, err := r1cs.Solve(fullWitness, solverOpts...)
	if err != nil {
		return nil, err
	}
	solution := _solution.(*cs.R1CSSolution)
	W, a, b, c := []fr.Element(solution.W), []fr.Element(solution.A), []fr.Element(solution.B), []fr.Element(solution.C)

	// CommitmentPok as prove.go builds it: challenge = fr.Hash(commitment WIRE VALUES, "G16-BSB22"), then
	// pedersen.BatchProve = sum_i challenge^i * ProveKnowledge_i.  The hash stays here; the MSMs over BasisExpSigma and the fold
	// ride in the proof's pool job (mi_prover_submit_bsb22), beside its five MSMs.
	var challenge fr.Element
	if len(commitmentInfo) > 0 {
		commitmentsSerialized := make([]byte, fr.Bytes*len(commitmentInfo))
		for i := range commitmentInfo {
			copy(commitmentsSerialized[fr.Bytes*i:], W[commitmentInfo[i].CommitmentIndex].Marshal())
		}
		ch, err := fr.Hash(commitmentsSerialized, []byte("G16-BSB22"), 1)
		if err != nil {
			return nil, err
		}
		challenge = ch[0]
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
	// out and pok live in C memory: the library writes them from a worker thread after this cgo call has returned
	out := (*C.mi_proof_out)(C.calloc(1, C.size_t(unsafe.Sizeof(C.mi_proof_out{}))))
	pok := (*C.mi_g1_affine)(C.calloc(1, C.size_t(unsafe.Sizeof(C.mi_g1_affine{}))))
	defer C.free(unsafe.Pointer(out))
	defer C.free(unsafe.Pointer(pok))
	var pin runtime.Pinner // W, a, b and the committed values are read by library threads until mi_prover_wait returns
	if len(W) == 0 || len(a) == 0 {
		return nil, errors.New("mi355x: empty solution")
	}
	pin.Pin(&W[0]); pin.Pin(&a[0]); pin.Pin(&b[0])
	defer pin.Unpin()
	// the mi_bsb22_input array lives in C memory too (it holds pointers to pinned Go arrays: the withKeyDesc rule)
	nb := 0
	for i := range privateCommittedValues {
		if len(privateCommittedValues[i]) > 0 {
			nb++
		}
	}
	var bsb *C.mi_bsb22_input
	if nb > 0 {
		if nb != len(commitmentInfo) {
			return nil, errors.New("mi355x: empty commitments between non-empty ones are not supported") // (gnark: every commitment has committed wires)
		}
		bsb = (*C.mi_bsb22_input)(C.calloc(C.size_t(nb), C.size_t(unsafe.Sizeof(C.mi_bsb22_input{}))))
		defer C.free(unsafe.Pointer(bsb))
		arr := unsafe.Slice(bsb, nb)
		pk.mu.Lock()
		for i := range privateCommittedValues {
			ppk, err := pk.pedersenKey(i)
			if err != nil {
				pk.mu.Unlock()
				return nil, err
			}
			pin.Pin(&privateCommittedValues[i][0])
			arr[i].key, arr[i].values, arr[i].n = ppk, (*C.mi_fr)(unsafe.Pointer(&privateCommittedValues[i][0])), C.size_t(len(privateCommittedValues[i]))
		}
		pk.mu.Unlock()
	}
	// c is NOT passed: solution.C = solution.A o solution.B row by row for every witness the solver accepts, and the library forms it
	// on the device (a quarter of the proof's PCIe bytes; mi355x_groth16.h, mi_groth16_prove).  `c` stays referenced for callers that
	// set PassC (e.g. a test that feeds an unsatisfied system on purpose).
	var cptr *C.mi_fr
	if PassC {
		pin.Pin(&c[0])
		cptr = (*C.mi_fr)(unsafe.Pointer(&c[0]))
	}
	var ticket C.uint64_t
	rc := C.mi_prover_submit_bsb22(pk.pool, pk.dev,
		(*C.mi_fr)(unsafe.Pointer(&W[0])), C.size_t(len(W)),
		(*C.mi_fr)(unsafe.Pointer(&a[0])), (*C.mi_fr)(unsafe.Pointer(&b[0])), cptr, C.size_t(len(a)),
		(*C.mi_fr)(unsafe.Pointer(&r)), (*C.mi_fr)(unsafe.Pointer(&s)), bsb, C.uint32_t(nb), (*C.mi_fr)(unsafe.Pointer(&challenge)),
		out, pok, nil, &ticket)
	if rc != C.MI_OK {
		return nil, fmt.Errorf("mi355x: mi_prover_submit_bsb22 rc=%d", int(rc))
	}
	if rc := C.mi_prover_wait(pk.pool, ticket); rc != C.MI_OK {
		return nil, fmt.Errorf("mi355x: rc=%d: %s", int(rc), C.GoString(C.mi_prover_last_error(pk.pool)))
	}
	runtime.KeepAlive(W)
	runtime.KeepAlive(c)
	proof.Ar = *(*bn254.G1Affine)(unsafe.Pointer(&out.ar))
	proof.Bs = *(*bn254.G2Affine)(unsafe.Pointer(&out.bs))
	proof.Krs = *(*bn254.G1Affine)(unsafe.Pointer(&out.krs))
	if nb > 0 {
		proof.CommitmentPok = *(*bn254.G1Affine)(unsafe.Pointer(pok))
	}
	return proof, nil
}

// PassC makes Prove upload solution.C as well instead of letting the device form it as A o B (the default; identical proofs for every
// witness the solver accepts).
var PassC = false

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

// batchProve is pedersen.BatchProve (gnark-crypto ecc/bn254/fr/pedersen) with the MSMs on the device:
// sum_i challenge^i * (sum_j values[i][j] * BasisExpSigma_i[j]).
func (pk *ProvingKey) batchProve(values [][]fr.Element, challenge fr.Element) (bn254.G1Affine, error) {
	poks := make([]bn254.G1Affine, len(values))
	pk.mu.Lock()
	defer pk.mu.Unlock()
	for i := range values {
		if len(values[i]) == 0 {
			continue // infinity
		}
		ppk, err := pk.pedersenKey(i)
		if err != nil {
			return bn254.G1Affine{}, err
		}
		if err := status(pk.ctx, C.mi_pedersen_prove_knowledge(pk.ctx, ppk, (*C.mi_fr)(unsafe.Pointer(&values[i][0])), C.size_t(len(values[i])),
			(*C.mi_g1_affine)(unsafe.Pointer(&poks[i])))); err != nil {
			return bn254.G1Affine{}, err
		}
	}
	var out bn254.G1Affine
	if len(poks) == 0 {
		return out, nil
	}
	if rc := C.mi_pedersen_fold((*C.mi_g1_affine)(unsafe.Pointer(&poks[0])), C.size_t(len(poks)), (*C.mi_fr)(unsafe.Pointer(&challenge)),
		(*C.mi_g1_affine)(unsafe.Pointer(&out))); rc != C.MI_OK {
		return out, fmt.Errorf("mi355x: mi_pedersen_fold rc=%d", int(rc))
	}
	return out, nil
}

// TraceRanges switches the library's roctx ranges on or off (mi_set_trace_ranges): the host-side phases of every call show up in
// `rocprofv3 --marker-trace --kernel-trace` beside the kernels.  Process-wide; an error when no roctx library can be loaded.
func TraceRanges(on bool) error {
	v := C.int32_t(0)
	if on {
		v = 1
	}
	if rc := C.mi_set_trace_ranges(v); rc != C.MI_OK {
		return fmt.Errorf("mi355x: trace ranges: rc=%d (no roctx library?)", int(rc))
	}
	return nil
}

func log2(n uint64) int { k := 0; for (uint64(1) << k) < n { k++ }; return k }

// g1 / g2: address of a point slice's backing array for a descriptor in C memory; the array is pinned until pin.Unpin
func g1(pin *runtime.Pinner, s []bn254.G1Affine) *C.mi_g1_affine {
	if len(s) == 0 {
		return nil
	}
	pin.Pin(&s[0])
	return (*C.mi_g1_affine)(unsafe.Pointer(&s[0]))
}
func g2(pin *runtime.Pinner, s []bn254.G2Affine) *C.mi_g2_affine {
	if len(s) == 0 {
		return nil
	}
	pin.Pin(&s[0])
	return (*C.mi_g2_affine)(unsafe.Pointer(&s[0]))
}

var _ = hash_to_field.New
