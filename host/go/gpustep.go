// The reference's frame loop (main.go:80-190) on the GPU engine.
//
// In main(), after the bodies exist (main.go:32-47) and the frame workers are started (main.go:50-61), replace
// the loop `for frame := startFrame; frame <= frames; frame++ { ... }` by
//
//	lastFrame = gpuRun(bodies, *tree, *nocollision, *norender, theta, dt, iterPerFrame, startFrame, frames, ch, lastFrame)
//
// and leave what follows it (close(ch), wg.Wait(), "Done. Took", exportFrameData) as it is.
package main

import (
	"flag"
	"fmt"
	"time"
)

// Flags for the engine's choices the reference has none for; registered beside the reference's own
// (main.go:17-24) and parsed by its flag.Parse().  The default precision is float64, the reference's arithmetic.
var (
	gpuPrecision = flag.String("precision", "f64", "GPU arithmetic: f64, mixed (fp32 pair terms, fp64 sums and state) or f32")
	gpuAsGo      = flag.Bool("asgo", false, "tree mode: the reference's running cell centres and unwalked last n mod 4 bodies")
	gpuPerWave   = flag.Bool("perwave", false, "tree mode: the bodies of a wavefront open cells together (not the reference's interaction set)")
	gpuCount     = flag.Int("gpus", 1, "GPUs driven by the one engine (with collisions on: -tree only)")
	gpuBlocking  = flag.Bool("blockingframes", false, "take frames with a blocking download instead of the side-stream ring")
	gpuDist      = flag.Bool("distributed", false, "with -gpus N -tree: every body on one GPU only, pruned subtrees exchanged")
)

func gpuOptionsFromFlags() gpuOptions {
	opt := gpuOptions{precision: int(nbePrecF64), asGo: *gpuAsGo, perWave: *gpuPerWave, gpus: *gpuCount, distributed: *gpuDist}
	switch *gpuPrecision {
	case "f32":
		opt.precision = int(nbePrecF32)
	case "mixed":
		opt.precision = int(nbePrecMixed)
	case "f64":
	default:
		panic("unknown -precision " + *gpuPrecision)
	}
	return opt
}

// gpuRun runs frames startFrame..frames (frames+1-startFrame iterations, as main.go:80 does): every iteration sends
// the state before its step to the workers unless norender (main.go:90-101), advances iterPerFrame steps
// (main.go:104-178) and prints the progress line (main.go:181-189).  Returns the last frame sent (what -save
// writes, main.go:199-201), or the incoming lastFrame when nothing was rendered.
func gpuRun(bodies []*body, tree, nocollision, norender bool, theta float64, dt float64, iterPerFrame int,
	startFrame, frames int, ch chan *frameJob, lastFrame *frameJob) *frameJob {
	eng := newGPUEngine(tree, nocollision, theta, dt, gpuOptionsFromFlags())
	defer eng.close()
	eng.upload(bodies)
	// Frames leave the GPU on a side stream into a ring of pinned buffers while the next step computes, and the
	// engine's delivery thread sends them on ch (framestream.go); -blockingframes keeps the synchronous copy.
	var fs *frameStream
	if !norender && !*gpuBlocking {
		fs = eng.streamFrames(iterPerFrame, cap(ch), startFrame, ch)
	}
	start := time.Now()
	for frame := startFrame; frame <= frames; frame++ {
		remainingBodies := 0
		if norender || fs != nil {
			remainingBodies = eng.alive()
		} else {
			lastFrame = eng.frame(frame)
			remainingBodies = len(lastFrame.Bodies)
			ch <- lastFrame
		}
		eng.step(iterPerFrame) // with streaming on, the copy of the state before this step goes out first

		avgTimePerFrame := time.Since(start).Milliseconds() / int64(frame-startFrame+1)
		done := 100.0
		if frames > startFrame {
			done = 100 * float64(frame-startFrame) / float64(frames-startFrame)
		}
		fmt.Printf("\r%.1f%%, %d bodies, %dms/frame, %s remaining, %s elapsed     ",
			done, remainingBodies, avgTimePerFrame,
			(time.Duration(avgTimePerFrame*int64(frames-frame)) * time.Millisecond).Round(time.Second),
			time.Since(start).Round(time.Second))
	}
	if fs != nil {
		eng.endFrames() // everything pending is delivered before the caller closes ch (main.go:191)
		if f := fs.lastFrame(); f != nil {
			lastFrame = f
		}
	}
	return lastFrame
}
