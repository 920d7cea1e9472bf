// Frame streaming: the copy of live bodies that the reference sends to its output workers before every step
// (main.go:90-101, channel of capacity 32 at main.go:50) leaves the GPU on a side stream into pinned host
// buffers while the step computes, and a delivery thread calls the host's callback.
#include "engine_internal.h"

__global__ __launch_bounds__(256) void narrow_state_kernel(const double4* __restrict__ posm, float4* __restrict__ out, int n)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const double4 p = posm[i];
    out[i] = make_float4((float)p.x, (float)p.y, (float)p.z, (float)p.w);
}

__global__ __launch_bounds__(256) void narrow_radius_kernel(const double* __restrict__ radius, float* __restrict__ out, int n)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = (float)radius[i];
}

static void delivery_loop(nbe_engine* e)
{
    FrameStream& fs = e->fs;
    (void)hipSetDevice(e->cfg.device);
    for (;;) {
        int slot;
        {
            std::unique_lock<std::mutex> lk(fs.mu);
            fs.cv_work.wait(lk, [&] { return fs.stop || !fs.queue.empty(); });
            if (fs.queue.empty()) return;            // stop requested and nothing left to deliver
            slot = fs.queue.front();
            fs.queue.pop_front();
        }
        FrameSlot& s = fs.slots[slot];
        (void)hipEventSynchronize(s.done);
        const int na_raw = *s.n_alive;               // -1: no tree build has removed anything yet
        size_t na = na_raw < 0 ? e->n : (size_t)na_raw;
        if (na > e->n) na = e->n;
        if (e->cfg.collisions && e->cfg.mode == NBE_MODE_DIRECT) {
            // direct sum: absorbed bodies stay in the arrays and are flagged; leave them out (main.go:91-95)
            size_t k = 0;
            for (size_t d = 0; d < na; d++) {
                if (s.dead[d]) continue;
                if (k != d) {
                    for (int c = 0; c < 4; c++) s.xyzm[4 * k + c] = s.xyzm[4 * d + c];
                    s.radius[k] = s.radius[d];
                    s.orig[k] = s.orig[d];
                }
                k++;
            }
            na = k;
        } else if (e->cfg.collisions) {
            // bodies absorbed by a collision since the last build are parked outside the bound without mass:
            // leave them out, as the nil check at main.go:91-95 does
            // (the parking position, in the very expression collide_and_update forms it with, so that both round alike
            // for any bound; it lies outside the bound, where no live body of a built tree can be)
            const double W = e->cfg.bound_width;
            const float px = (float)((e->cfg.bound_center[0] - 0.5 * W) + 2.0 * W), py = (float)((e->cfg.bound_center[1] - 0.5 * W) + 2.0 * W),
                        pz = (float)((e->cfg.bound_center[2] - 0.5 * W) + 2.0 * W);
            size_t k = 0;
            for (size_t d = 0; d < na; d++) {
                const float* r = s.xyzm + 4 * d;
                if (r[3] == 0.0f && r[0] == px && r[1] == py && r[2] == pz) continue;
                if (k != d) {
                    for (int c = 0; c < 4; c++) s.xyzm[4 * k + c] = r[c];
                    s.radius[k] = s.radius[d];
                    s.orig[k] = s.orig[d];
                }
                k++;
            }
            na = k;
        }
        fs.ids_scratch.resize(na);
        for (size_t d = 0; d < na; d++) fs.ids_scratch[d] = e->ids[s.orig[d]];
        fs.cb(fs.user, s.frame, na, s.xyzm, e->cfg.collisions ? s.radius : nullptr, fs.ids_scratch.data());
        {
            std::lock_guard<std::mutex> lk(fs.mu);
            s.busy = false;
        }
        fs.cv_free.notify_all();
    }
}

// Frees whatever frame_stream_allocate set up (every pointer is tested, so a partial set-up is fine).
static void frame_stream_release(nbe_engine* e)
{
    FrameStream& fs = e->fs;
    (void)hipSetDevice(e->cfg.device);
    if (fs.side) (void)hipStreamSynchronize(fs.side);
    for (FrameSlot& s : fs.slots) {
        if (s.xyzm) (void)hipHostFree(s.xyzm);
        if (s.orig) (void)hipHostFree(s.orig);
        if (s.radius) (void)hipHostFree(s.radius);
        if (s.dead) (void)hipHostFree(s.dead);
        if (s.n_alive) (void)hipHostFree(s.n_alive);
        if (s.d_n_alive) (void)hipFree(s.d_n_alive);
        if (s.done) (void)hipEventDestroy(s.done);
    }
    fs.slots.clear();
    if (fs.staging) { (void)hipFree(fs.staging); fs.staging = nullptr; }
    if (fs.radius_staging) { (void)hipFree(fs.radius_staging); fs.radius_staging = nullptr; }
    if (fs.state_ready) { (void)hipEventDestroy(fs.state_ready); fs.state_ready = nullptr; }
    for (int b = 0; b < 2; b++)
        if (fs.copy_done[b]) { (void)hipEventDestroy(fs.copy_done[b]); fs.copy_done[b] = nullptr; }
    if (fs.side) { (void)hipStreamDestroy(fs.side); fs.side = nullptr; }
    fs.copy_pending[0] = fs.copy_pending[1] = false;
}

static int frame_stream_allocate(nbe_engine* e, int slots)
{
    FrameStream& fs = e->fs;
    const size_t cap = e->n > 0 ? e->n : 1;
    NBE_HIP(hipStreamCreateWithFlags(&fs.side, hipStreamNonBlocking));
    NBE_HIP(hipEventCreateWithFlags(&fs.state_ready, hipEventDisableTiming));
    NBE_HIP(hipEventCreateWithFlags(&fs.copy_done[0], hipEventDisableTiming));
    NBE_HIP(hipEventCreateWithFlags(&fs.copy_done[1], hipEventDisableTiming));
    if (e->state64) NBE_HIP(hipMalloc((void**)&fs.staging, cap * sizeof(float4)));
    if (e->cfg.collisions) NBE_HIP(hipMalloc((void**)&fs.radius_staging, cap * sizeof(float)));
    fs.slots.assign((size_t)slots, FrameSlot());
    for (FrameSlot& s : fs.slots) {
        NBE_HIP(hipHostMalloc((void**)&s.xyzm, cap * 4 * sizeof(float), hipHostMallocDefault));
        NBE_HIP(hipHostMalloc((void**)&s.orig, cap * sizeof(uint32_t), hipHostMallocDefault));
        if (e->cfg.collisions) NBE_HIP(hipHostMalloc((void**)&s.radius, cap * sizeof(float), hipHostMallocDefault));
        if (e->cfg.collisions && e->cfg.mode == NBE_MODE_DIRECT) NBE_HIP(hipHostMalloc((void**)&s.dead, cap, hipHostMallocDefault));
        NBE_HIP(hipHostMalloc((void**)&s.n_alive, sizeof(int), hipHostMallocDefault));
        NBE_HIP(hipMalloc((void**)&s.d_n_alive, sizeof(int)));
        NBE_HIP(hipEventCreateWithFlags(&s.done, hipEventDisableTiming));
    }
    return NBE_OK;
}

extern "C" int nbe_frame_stream_begin(nbe_engine* e, int every, int slots, int first_frame, nbe_frame_cb cb, void* user)
{
    if (!e || !cb) { nbe_set_error("nbe_frame_stream_begin: NULL argument"); return NBE_E_INVALID; }
    if (multi_is_distributed(e)) return multi_dist_frames_begin(e, every, slots, first_frame, cb, user);      // every GPU sends its own bodies
    if (e->multi) e = multi_first(e);          // several GPUs: every one holds the full state; the first one streams it
    if (every < 1 || slots < 1 || slots > 1024) { nbe_set_error("every must be >= 1 and slots in 1..1024"); return NBE_E_INVALID; }
    if (e->fs.active) { nbe_set_error("frame streaming is already on"); return NBE_E_INVALID; }
    NBE_HIP(hipSetDevice(e->cfg.device));
    FrameStream& fs = e->fs;
    fs.every = every; fs.first_frame = first_frame; fs.cb = cb; fs.user = user;
    fs.step_base = e->stats.steps;
    fs.have_last = false;
    fs.stop = false; fs.copy_pending[0] = fs.copy_pending[1] = false;
    fs.queue.clear();
    int rc;
    try {
        rc = frame_stream_allocate(e, slots);
        if (rc == NBE_OK) {
            fs.active = true;
            fs.worker = std::thread(delivery_loop, e);
        }
    } catch (const std::exception& ex) {
        nbe_set_error("frame streaming could not start: %s", ex.what());
        rc = NBE_E_NOMEM;
    }
    if (rc != NBE_OK) {                            // keep the message of the failure, drop what was set up
        fs.active = false;
        frame_stream_release(e);
    }
    return rc;
}

// Called before any kernel that writes state buffer `buf`: if a frame copy is still reading that buffer, the
// compute stream waits for it.  (Direct mode writes the other buffer during the overlapping step, so the copy
// hides behind the whole force pass; tree mode writes the copied buffer again in its force pass, so the copy
// hides behind the sort and build.)
int frame_stream_guard_write(nbe_engine* e, int buf)
{
    FrameStream& fs = e->fs;
    if (fs.active && fs.copy_pending[buf]) {
        NBE_HIP(hipStreamWaitEvent(e->stream, fs.copy_done[buf], 0));
        fs.copy_pending[buf] = false;
    }
    return NBE_OK;
}

// Called at the top of every step with the state that the step is about to read in e->posm[e->cur].
int frame_stream_before_step(nbe_engine* e)
{
    FrameStream& fs = e->fs;
    if (!fs.active) return NBE_OK;
    const uint64_t k = e->stats.steps - fs.step_base;
    if (k % (uint64_t)fs.every != 0 || e->n == 0) return NBE_OK;
    if (fs.have_last && fs.last_k == k) return NBE_OK;    // this step's frame is out already (nbe_shard_prepare, then nbe_step)
    fs.have_last = true; fs.last_k = k;
    int slot = -1;
    {
        std::unique_lock<std::mutex> lk(fs.mu);      // back-pressure: wait for a free buffer like a full channel
        fs.cv_free.wait(lk, [&] {
            for (size_t i = 0; i < fs.slots.size(); i++)
                if (!fs.slots[i].busy) { slot = (int)i; return true; }
            return false;
        });
        fs.slots[slot].busy = true;
    }
    FrameSlot& s = fs.slots[slot];
    s.frame = fs.first_frame + (int)(k / (uint64_t)fs.every);
    const size_t n = e->n;
    // the build of the coming step resets the live count on the compute stream: snapshot it there first
    if (e->have_counts)
        NBE_HIP(hipMemcpyAsync(s.d_n_alive, e->tw.counters, sizeof(int), hipMemcpyDeviceToDevice, e->stream));
    NBE_HIP(hipEventRecord(fs.state_ready, e->stream));
    NBE_HIP(hipStreamWaitEvent(fs.side, fs.state_ready, 0));
    const void* src = e->posm[e->cur];
    if (e->state64) {
        hipLaunchKernelGGL(narrow_state_kernel, dim3(cdiv(n, 256)), dim3(256), 0, fs.side, (const double4*)e->posm[e->cur],
                           fs.staging, (int)n);
        NBE_HIP(hipGetLastError());
        src = fs.staging;
    }
    NBE_HIP(hipMemcpyAsync(s.xyzm, src, n * sizeof(float4), hipMemcpyDeviceToHost, fs.side));
    NBE_HIP(hipMemcpyAsync(s.orig, e->orig[e->orig_cur], n * sizeof(uint32_t), hipMemcpyDeviceToHost, fs.side));
    if (e->cfg.collisions) {
        // the radii live beside the state buffer and are rewritten together with it, so the same guard covers them
        const bool direct = e->cfg.mode == NBE_MODE_DIRECT;
        hipLaunchKernelGGL(narrow_radius_kernel, dim3(cdiv(n, 256)), dim3(256), 0, fs.side,
                           (const double*)(direct ? e->radius[0] : e->radius[e->cur]), fs.radius_staging, (int)n);
        // direct sum: radii and flags are single arrays that only a replayed step writes; that step waits for this
        // copy through the guard on the state buffer it writes next
        if (direct) NBE_HIP(hipMemcpyAsync(s.dead, e->cw.dead, n, hipMemcpyDeviceToHost, fs.side));
        NBE_HIP(hipGetLastError());
        NBE_HIP(hipMemcpyAsync(s.radius, fs.radius_staging, n * sizeof(float), hipMemcpyDeviceToHost, fs.side));
    }
    if (e->have_counts) NBE_HIP(hipMemcpyAsync(s.n_alive, s.d_n_alive, sizeof(int), hipMemcpyDeviceToHost, fs.side));
    else *s.n_alive = -1;
    NBE_HIP(hipEventRecord(s.done, fs.side));
    NBE_HIP(hipEventRecord(fs.copy_done[e->cur], fs.side));
    fs.copy_pending[e->cur] = true;
    {
        std::lock_guard<std::mutex> lk(fs.mu);
        fs.queue.push_back(slot);
    }
    fs.cv_work.notify_one();
    return NBE_OK;
}

int frame_stream_stop(nbe_engine* e)
{
    FrameStream& fs = e->fs;
    if (!fs.active) return NBE_OK;
    {
        std::lock_guard<std::mutex> lk(fs.mu);
        fs.stop = true;
    }
    fs.cv_work.notify_all();
    if (fs.worker.joinable()) fs.worker.join();      // delivers everything still queued first
    frame_stream_release(e);
    fs.active = false;
    return NBE_OK;
}

extern "C" int nbe_frame_stream_end(nbe_engine* e)
{
    if (!e) { nbe_set_error("engine is NULL"); return NBE_E_INVALID; }
    if (multi_is_distributed(e)) return multi_dist_frames_end(e);
    if (e->multi) e = multi_first(e);
    return frame_stream_stop(e);
}
