// cgo binding of libnbody_engine.so for the reference program (package main of quillaja/nbody).
//
// Drop this file and gpustep.go next to the reference's sources, with include/ and nbody_amd/ of this
// repository beside them (or adjust the #cgo lines), and call gpuRun from main in place of the frame loop
// (main.go:80-190).  Everything else of the reference stays as it is: flags, makebodies, the frame workers
// (render.go, database.go, gob.go), importFrameData / exportFrameData.
//
// No Go toolchain exists in the image this engine was built in, so these two files have not been compiled;
// host/go/check.sh builds them where `go` is available.
package main

/*
#cgo CFLAGS: -I${SRCDIR}/include
#cgo LDFLAGS: -L${SRCDIR}/nbody_amd -lnbody_engine -Wl,-rpath,${SRCDIR}/nbody_amd
#include <stdlib.h>
#include "nbody_engine.h"
*/
import "C"

import (
	"runtime"
	"unsafe"
)

// The precision constants as Go values (gpustep.go has no cgo preamble of its own).
const (
	nbePrecF32   = C.NBE_PREC_F32
	nbePrecMixed = C.NBE_PREC_MIXED
	nbePrecF64   = C.NBE_PREC_F64
)

// gpuOptions are the engine's choices the reference has no flag for.
type gpuOptions struct {
	precision int  // C.NBE_PREC_F64 (the reference computes in float64; gpuOptionsFromFlags' default), C.NBE_PREC_MIXED, C.NBE_PREC_F32
	asGo      bool // tree mode: running cell centres (tree.go:189-193) and unwalked last n mod 4 bodies (main.go:134-148)
	perWave   bool // tree mode: the bodies of a wavefront open cells together (faster, finer, not the reference's set)
	gpus      int  // > 1: the engine drives GPUs 0 .. gpus-1 itself (body ranges fanned out as main.go:134-149 fans them over goroutines); with collisions on, tree mode only
	distributed bool // with gpus > 1: the key-range-distributed tree (no GPU holds all bodies)
}

// gpuEngine owns one nbe_engine.  All calls happen on one locked OS thread, as the reference drives the step
// from main only (main.go:80,104).  No Go pointer is retained by C after a call returns.
type gpuEngine struct {
	h       *C.nbe_engine
	n       int // bodies uploaded
	pos     []float64 // xyzxyz..., reused between calls
	vel     []float64
	m       []float64
	rad     []float64
	id      []uint64
	radByID map[uint64]float64 // radii of an engine without collisions (they never change there)
	collide bool
	streamToken uintptr // key of this engine's frame stream (framestream.go)
}

func gpuCheck(rc C.int) {
	if rc != 0 { // the reference panics on every error (main.go:213-238)
		panic(C.GoString(C.nbe_last_error()))
	}
}

// newGPUEngine: tree = the -tree flag (main.go:22), nocollision = main.go:24, theta = main.go:73, dt = main.go:28.
func newGPUEngine(tree, nocollision bool, theta, dt float64, opt gpuOptions) *gpuEngine {
	runtime.LockOSThread()
	var cfg C.nbe_config
	gpuCheck(C.nbe_default_config(&cfg)) // G (physics.go:132), theta 1, dt 3600, bound 2^16 (main.go:74-76)
	if tree {
		cfg.mode = C.int(C.NBE_MODE_TREE)
	}
	cfg.precision = C.int(opt.precision)
	cfg.theta = C.double(theta)
	cfg.dt = C.double(dt)
	if !nocollision {
		cfg.collisions = 1
	}
	if opt.gpus > 1 {
		cfg.num_devices = C.int(opt.gpus) // device_ids stays nil: devices 0 .. gpus-1
		if opt.distributed {
			cfg.distributed = 1 // every body on one GPU only; needs -tree
		}
	}
	if tree && opt.asGo {
		cfg.reference_quirks = C.int(C.NBE_QUIRK_REF_COM | C.NBE_QUIRK_DROP_TAIL)
	} else if tree && opt.perWave {
		cfg.walk_criterion = C.int(C.NBE_WALK_PER_WAVE)
	}
	e := &gpuEngine{collide: !nocollision}
	gpuCheck(C.nbe_create(&cfg, &e.h))
	return e
}

// upload replaces the slice of makebodies / overwriteBodies (physics.go:16-73, main.go:243-248).  nil entries
// (bodies removed earlier) are left out; insertion order and "n mod 4" of the asGo mode refer to what is uploaded.
func (e *gpuEngine) upload(bodies []*body) {
	e.n = 0
	for _, b := range bodies {
		if b != nil {
			e.n++
		}
	}
	e.pos, e.vel = make([]float64, 3*e.n), make([]float64, 3*e.n)
	e.m, e.rad, e.id = make([]float64, e.n), make([]float64, e.n), make([]uint64, e.n)
	e.radByID = make(map[uint64]float64, e.n)
	k := 0
	for _, b := range bodies {
		if b == nil {
			continue
		}
		e.pos[3*k], e.pos[3*k+1], e.pos[3*k+2] = b.X, b.Y, b.Z
		e.vel[3*k], e.vel[3*k+1], e.vel[3*k+2] = b.Vx, b.Vy, b.Vz
		e.m[k], e.rad[k], e.id[k] = b.Mass, b.Radius, b.ID
		e.radByID[b.ID] = b.Radius
		k++
	}
	if e.n == 0 {
		gpuCheck(C.nbe_upload(e.h, 0, nil, nil, nil, nil))
		return
	}
	gpuCheck(C.nbe_upload(e.h, C.size_t(e.n),
		(*C.double)(unsafe.Pointer(&e.pos[0])), (*C.double)(unsafe.Pointer(&e.vel[0])),
		(*C.double)(unsafe.Pointer(&e.m[0])), (*C.uint64_t)(unsafe.Pointer(&e.id[0]))))
	if e.collide {
		gpuCheck(C.nbe_set_radius(e.h, (*C.double)(unsafe.Pointer(&e.rad[0]))))
	}
}

// step replaces iters iterations of main.go:104-178.
func (e *gpuEngine) step(iters int) { gpuCheck(C.nbe_step(e.h, C.int(iters))) }

// alive is remainingBodies of the progress line (main.go:83-88,181-189).
func (e *gpuEngine) alive() int {
	var st C.nbe_stats_t
	gpuCheck(C.nbe_stats(e.h, &st))
	return int(st.n_alive)
}

// frame replaces the copy at main.go:90-101: live bodies only (bodies removed by tree.go:290-292 or absorbed by
// physics.go:153-162 are left out the way the nil check at main.go:91-95 leaves them out), upload order.
func (e *gpuEngine) frame(frame int) *frameJob {
	var alive C.size_t
	if e.n > 0 {
		gpuCheck(C.nbe_download(e.h,
			(*C.double)(unsafe.Pointer(&e.pos[0])), (*C.double)(unsafe.Pointer(&e.vel[0])),
			(*C.double)(unsafe.Pointer(&e.m[0])), (*C.uint64_t)(unsafe.Pointer(&e.id[0])), &alive))
		if e.collide { // merged radii, in the order of the download
			gpuCheck(C.nbe_get_radius(e.h, (*C.double)(unsafe.Pointer(&e.rad[0])), nil))
		}
	}
	bs := make([]body, int(alive))
	for k := range bs {
		r := 0.0
		if e.collide {
			r = e.rad[k]
		} else {
			r = e.radByID[e.id[k]]
		}
		bs[k] = body{ID: e.id[k], Mass: e.m[k], Radius: r,
			X: e.pos[3*k], Y: e.pos[3*k+1], Z: e.pos[3*k+2],
			Vx: e.vel[3*k], Vy: e.vel[3*k+1], Vz: e.vel[3*k+2]}
	}
	return &frameJob{Frame: frame, Bodies: bs} // render.go:33-36
}

func (e *gpuEngine) close() {
	C.nbe_destroy(e.h)
	runtime.UnlockOSThread()
}
