#!/bin/bash
# Builds the reference program with the GPU binding where a Go toolchain exists:
#   host/go/check.sh /path/to/quillaja-nbody-checkout
# Copies engine.go, gpustep.go, framestream.go and frame_bridge.c, the header and the engine library next to the reference's sources (in a scratch
# copy of the checkout), and runs `go vet` and `go build`.  The reference's own main.go still has to call gpuRun
# (see gpustep.go); until then the build only proves that the binding compiles against the reference's types.
set -eu
command -v go >/dev/null || { echo "no go toolchain: skipped"; exit 0; }
REF=${1:?path to a checkout of the reference}
HERE=$(cd "$(dirname "$0")" && pwd); ROOT=$(cd "$HERE/../.." && pwd)
W=$(mktemp -d)
cp -r "$REF"/. "$W"/
cp "$HERE"/engine.go "$HERE"/gpustep.go "$HERE"/framestream.go "$HERE"/frame_bridge.c "$W"/
mkdir -p "$W/include" "$W/nbody_amd"
cp "$ROOT/include/nbody_engine.h" "$W/include/"
cp "$ROOT/nbody_amd/libnbody_engine.so" "$W/nbody_amd/"
cd "$W" && go vet ./... && go build -o nbody_gpu . && echo "built $W/nbody_gpu"
