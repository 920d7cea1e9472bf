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
	"fmt"
	"time"
)

// gpuRun runs frames startFrame..frames (frames+1-startFrame iterations, as main.go:80 does): every iteration sends
// the state before its step to the workers unless norender (main.go:90-101), advances iterPerFrame steps
// (main.go:104-178) and prints the progress line (main.go:181-189).  Returns the last frame sent (what -save
// writes, main.go:199-201), or the incoming lastFrame when nothing was rendered.
func gpuRun(bodies []*body, tree, nocollision, norender bool, theta float64, dt float64, iterPerFrame int,
	startFrame, frames int, ch chan *frameJob, lastFrame *frameJob) *frameJob {
	eng := newGPUEngine(tree, nocollision, theta, dt, gpuOptions{})
	defer eng.close()
	eng.upload(bodies)
	start := time.Now()
	for frame := startFrame; frame <= frames; frame++ {
		remainingBodies := 0
		if norender {
			remainingBodies = eng.alive()
		} else {
			lastFrame = eng.frame(frame)
			remainingBodies = len(lastFrame.Bodies)
			ch <- lastFrame
		}
		eng.step(iterPerFrame)

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
	return lastFrame
}
