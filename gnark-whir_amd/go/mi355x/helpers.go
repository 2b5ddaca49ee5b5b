//go:build mi355x

package mi355x

// Helpers of the shim (SOURCE ONLY, never compiled here — see mi355x.go).

import "sort"

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

