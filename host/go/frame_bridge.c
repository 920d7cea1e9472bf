/* C side of the Go frame callback (see framestream.go): cgo cannot take the address of an exported Go function
 * in the file that exports it, so the call that passes it to the engine lives here. */
#include <stdint.h>
#include "nbody_engine.h"
#include "_cgo_export.h"

int nbeGoFrameStreamBegin(nbe_engine* e, int every, int slots, int first_frame, uintptr_t token)
{
    /* the token is an integer key of a Go-side table, carried in the user pointer; no Go pointer crosses */
    return nbe_frame_stream_begin(e, every, slots, first_frame, (nbe_frame_cb)goFrameCallback, (void*)token);
}
