// Frames through the engine's pinned ring instead of a blocking download.
//
// The reference copies the live bodies into a frameJob before every step and sends it to its output workers
// (main.go:90-101).  Here that copy leaves the GPU on a side stream while the step computes
// (nbe_frame_stream_begin); the engine's delivery thread calls goFrameCallback, which builds the frameJob and
// sends it on the same channel.  The channel send blocks the delivery thread when the workers fall behind, and
// the ring of pinned buffers then blocks nbe_step: the back-pressure of the reference's buffered channel.
//
// Not compiled where this was written (no Go toolchain); host/go/check.sh builds it where `go` exists.
package main

/*
#include <stdint.h>
#include "nbody_engine.h"

// defined in frame_bridge.c: starts streaming with the exported Go callback below
int nbeGoFrameStreamBegin(nbe_engine* e, int every, int slots, int first_frame, uintptr_t token);
*/
import "C"

import (
	"sync"
	"unsafe"
)

// frameStream is what the callback needs to know about one engine.
type frameStream struct {
	eng  *gpuEngine
	ch   chan *frameJob
	mu   sync.Mutex
	last *frameJob // lastFrame of main.go:97-100
}

// C keeps only an integer token, never a Go pointer (cgo rule); the token finds the stream here.
var (
	frameStreamsMu sync.Mutex
	frameStreams   = map[uintptr]*frameStream{}
	frameStreamSeq uintptr
)

// streamFrames starts the stream: frame k (numbered from startFrame) is the state before step k*iterPerFrame.
// slots is the capacity of the reference's channel (32, main.go:50).
func (e *gpuEngine) streamFrames(iterPerFrame, slots, startFrame int, ch chan *frameJob) *frameStream {
	fs := &frameStream{eng: e, ch: ch}
	frameStreamsMu.Lock()
	frameStreamSeq++
	token := frameStreamSeq
	frameStreams[token] = fs
	frameStreamsMu.Unlock()
	e.streamToken = token
	gpuCheck(C.nbeGoFrameStreamBegin(e.h, C.int(iterPerFrame), C.int(slots), C.int(startFrame), C.uintptr_t(token)))
	return fs
}

// endFrames delivers what is pending and stops the delivery thread (before close(ch), main.go:191).
func (e *gpuEngine) endFrames() {
	gpuCheck(C.nbe_frame_stream_end(e.h))
	frameStreamsMu.Lock()
	delete(frameStreams, e.streamToken)
	frameStreamsMu.Unlock()
}

func (fs *frameStream) lastFrame() *frameJob {
	fs.mu.Lock()
	defer fs.mu.Unlock()
	return fs.last
}

// goFrameCallback runs on the engine's delivery thread.  The arrays are valid only during the call, so they are
// copied into the frameJob (render.go:33-36) here.
//
//export goFrameCallback
func goFrameCallback(user unsafe.Pointer, frame C.int, n C.size_t, xyzm *C.float, radius *C.float, ids *C.uint64_t) {
	frameStreamsMu.Lock()
	fs := frameStreams[uintptr(user)]
	frameStreamsMu.Unlock()
	if fs == nil {
		return
	}
	count := int(n)
	job := &frameJob{Frame: int(frame), Bodies: make([]body, count)}
	if count > 0 {
		// (array-pointer slicing instead of unsafe.Slice: the reference's go.mod declares go 1.14)
		p := (*[1 << 30]float32)(unsafe.Pointer(xyzm))[: 4*count : 4*count]
		id := (*[1 << 28]uint64)(unsafe.Pointer(ids))[:count:count]
		var rad []float32
		if radius != nil { // collisions on: merged radii travel with the frame
			rad = (*[1 << 30]float32)(unsafe.Pointer(radius))[:count:count]
		}
		for k := 0; k < count; k++ {
			r := fs.eng.radByID[id[k]]
			if rad != nil {
				r = float64(rad[k])
			}
			job.Bodies[k] = body{ID: id[k], Mass: float64(p[4*k+3]), Radius: r,
				X: float64(p[4*k]), Y: float64(p[4*k+1]), Z: float64(p[4*k+2])}
		}
	}
	fs.mu.Lock()
	fs.last = job
	fs.mu.Unlock()
	fs.ch <- job // main.go:101
}
