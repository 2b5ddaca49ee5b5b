//go:build mi355x

package mi355x

// One groth16.Prove spread over several MI355X of a node (SURVEY 8e, BASELINE configs[4]) -- SOURCE ONLY, never compiled
// (no Go toolchain in the build image; see mi355x.go).  The key is point-sharded at load (slice r of pk.G1.{A,B,K,Z} / pk.G2.B
// stays resident on GPU r); per proof only scalars move; mode 0 combines one partial sum per MSM, mode 1 reduce-scatters
// the bucket sums over RCCL first (include/mi355x_groth16.h, device groups).  Proof bytes equal Prove's.

/*
#include "mi355x_groth16.h"
#include "mi355x_groth16_group.h"
*/
import "C"

import (
	"fmt"
	"unsafe"

	"github.com/consensys/gnark-crypto/ecc/bn254"
	"github.com/consensys/gnark-crypto/ecc/bn254/fr"
	groth16_bn254 "github.com/consensys/gnark/backend/groth16/bn254"
	cs "github.com/consensys/gnark/constraint/bn254"
)

// Group is a set of devices that prove together.
type Group struct {
	g   *C.mi_group
	spk *C.mi_pk_sharded
}

func NewGroup(devices []int) (*Group, error) {
	ids := make([]C.int, len(devices))
	for i, d := range devices {
		ids[i] = C.int(d)
	}
	var g *C.mi_group
	if rc := C.mi_group_create(&ids[0], C.int(len(ids)), &g); rc != C.MI_OK {
		return nil, fmt.Errorf("mi355x: mi_group_create rc=%d", int(rc))
	}
	return &Group{g: g}, nil
}

func (gr *Group) status(rc C.int32_t) error {
	if rc == C.MI_OK {
		return nil
	}
	return fmt.Errorf("mi355x: rc=%d: %s", int(rc), C.GoString(C.mi_group_last_error(gr.g)))
}

// SetLeadShare: rank 0 also runs computeH, so it takes permille / 1000 of an even share of the wires (mi_group_set_lead_share;
// LeadShareAuto = 1000 / 500 / 0 for 1 / 2 / >= 3 devices, the default).  Before LoadKey.
const LeadShareAuto = 0xffffffff

func (gr *Group) SetLeadShare(permille uint32) error {
	return gr.status(C.mi_group_set_lead_share(gr.g, C.uint32_t(permille)))
}

// ShardComputeH makes ProveSolved run computeH over all devices of the group (four-step transforms, mi_group_set_sharded_compute_h)
// instead of on the first one alone: 2, 4, 8 or 16 devices.  Same proofs.
func (gr *Group) ShardComputeH(on bool) error {
	v := C.uint32_t(0)
	if on {
		v = 1
	}
	return gr.status(C.mi_group_set_sharded_compute_h(gr.g, v))
}

// LoadKey shards pk over the group's devices.  The descriptor is built in C memory over pinned Go arrays (ProvingKey.withKeyDesc,
// mi355x.go: a Go struct holding Go pointers must not be passed to C by address).
func (gr *Group) LoadKey(pk *ProvingKey, r1cs *cs.R1CS) error {
	return pk.withKeyDesc(r1cs, func(d *C.mi_pk_desc) error {
		return gr.status(C.mi_pk_load_sharded(gr.g, d, &gr.spk))
	})
}

// ProveSolved runs the post-solve part of groth16.Prove over the group (W, a, b, c from r1cs.Solve; r, s sampled by the caller
// in prove.go's order).
func (gr *Group) ProveSolved(W, a, b, c []fr.Element, r, s *fr.Element, mode uint32) (*groth16_bn254.Proof, error) {
	if len(W) == 0 || len(a) == 0 || len(b) != len(a) || len(c) != len(a) {
		return nil, fmt.Errorf("mi355x: empty or ragged solution")
	}
	var out C.mi_proof_out // a Go value without pointers, written during the call only
	rc := C.mi_groth16_prove_sharded(gr.g, gr.spk,
		(*C.mi_fr)(unsafe.Pointer(&W[0])), C.size_t(len(W)),
		(*C.mi_fr)(unsafe.Pointer(&a[0])), (*C.mi_fr)(unsafe.Pointer(&b[0])), (*C.mi_fr)(unsafe.Pointer(&c[0])), C.size_t(len(a)),
		(*C.mi_fr)(unsafe.Pointer(r)), (*C.mi_fr)(unsafe.Pointer(s)), C.uint32_t(mode), &out, nil)
	if err := gr.status(rc); err != nil {
		return nil, err
	}
	p := &groth16_bn254.Proof{}
	p.Ar = *(*bn254.G1Affine)(unsafe.Pointer(&out.ar))
	p.Bs = *(*bn254.G2Affine)(unsafe.Pointer(&out.bs))
	p.Krs = *(*bn254.G1Affine)(unsafe.Pointer(&out.krs))
	return p, nil
}

func (gr *Group) Close() {
	if gr.spk != nil {
		C.mi_pk_sharded_free(gr.g, gr.spk)
	}
	if gr.g != nil {
		C.mi_group_destroy(gr.g)
	}
}
