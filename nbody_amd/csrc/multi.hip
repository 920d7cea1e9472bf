// Several GPUs behind one engine handle (nbe_config.num_devices > 1).
//
// The reference's only data-parallel strategy is the fan-out of main.go:134-149: contiguous ranges of the body
// slice go to workers that share one read-only tree.  Here a worker is a GPU.  The coordinator below owns one
// ordinary engine per GPU, all holding the full body set; per step every engine sorts and builds (tree mode; the
// build is deterministic, so all trees and sorted orders are identical), advances its own range into the
// next-state buffers (nbe_shard_advance), and then pulls the other engines' ranges over the peer links into its own
// next-state buffers.  All of it is queued on the engines' streams and ordered by events; the host never waits
// inside a step.
#include "engine_internal.h"

#include <new>
#include <vector>

struct MultiState {
    std::vector<nbe_engine*> subs;
    std::vector<int> devs;
    std::vector<uint64_t> first, count;      // range of device-order bodies each engine advances
    std::vector<hipEvent_t> advanced;        // engine r: its own range of *_next is written (recorded on its stream)
    std::vector<hipEvent_t> pulled;          // engine r: it has finished copying the other ranges (recorded on its stream)
    std::vector<hipStream_t> side;           // engine r: stream of its velocity copies (tree mode: they run beside the key sort)
    std::vector<hipEvent_t> vel_done;        // engine r: its velocity copies have landed (recorded on side[r])
};

nbe_engine* multi_first(nbe_engine* e) { return e->multi->subs[0]; }

static void ranges(size_t n, size_t parts, std::vector<uint64_t>* first, std::vector<uint64_t>* count)
{
    // Ranges start at multiples of 256 bodies, so that a wavefront of the tree walk holds the same 64 bodies as in a
    // one-GPU run: mixed precision folds its fp32 partial sums into the fp64 totals every 64 node visits of the
    // wavefront, and only then do the roundings fall where they fall on one GPU.
    size_t per = (n + parts - 1) / parts;
    per = (per + 255) / 256 * 256;
    first->assign(parts, 0);
    count->assign(parts, 0);
    for (size_t r = 0; r < parts; r++) {
        const size_t f = r * per < n ? r * per : n;
        (*first)[r] = f;
        (*count)[r] = per < n - f ? per : n - f;
    }
}

void multi_destroy(nbe_engine* e)
{
    MultiState* m = e->multi;
    if (m) {
        for (size_t r = 0; r < m->subs.size(); r++) {
            if (m->subs[r]) (void)nbe_sync(m->subs[r]);
        }
        for (size_t r = 0; r < m->subs.size(); r++) {
            (void)hipSetDevice(m->devs[r]);
            if (r < m->advanced.size() && m->advanced[r]) (void)hipEventDestroy(m->advanced[r]);
            if (r < m->pulled.size() && m->pulled[r]) (void)hipEventDestroy(m->pulled[r]);
            if (r < m->vel_done.size() && m->vel_done[r]) (void)hipEventDestroy(m->vel_done[r]);
            if (r < m->side.size() && m->side[r]) (void)hipStreamDestroy(m->side[r]);
            if (m->subs[r]) nbe_destroy(m->subs[r]);
        }
        delete m;
    }
    delete e;
}

int multi_create(const nbe_config* cfg, nbe_engine** out)
{
    const int G = cfg->num_devices;
    if (G > 64) { nbe_set_error("num_devices %d: at most 64", G); return NBE_E_INVALID; }
    if (cfg->collisions && cfg->mode != NBE_MODE_TREE) {
        nbe_set_error("collisions with num_devices > 1 need tree mode");
        return NBE_E_UNSUPPORTED;
    }
    nbe_engine* e = new (std::nothrow) nbe_engine();
    MultiState* m = new (std::nothrow) MultiState();
    if (!e || !m) { delete e; delete m; nbe_set_error("out of host memory"); return NBE_E_NOMEM; }
    e->multi = m;
    e->cfg = *cfg;
    e->cfg.device_ids = nullptr;             // the caller's array is not retained
    m->subs.assign((size_t)G, nullptr);
    m->advanced.assign((size_t)G, nullptr);
    m->pulled.assign((size_t)G, nullptr);
    m->side.assign((size_t)G, nullptr);
    m->vel_done.assign((size_t)G, nullptr);
    for (int r = 0; r < G; r++) m->devs.push_back(cfg->device_ids ? cfg->device_ids[r] : r);
    int rc = NBE_OK;
    for (int r = 0; r < G && rc == NBE_OK; r++) {
        nbe_config sub = *cfg;
        sub.num_devices = 1;
        sub.device_ids = nullptr;
        sub.device = m->devs[(size_t)r];
        rc = nbe_create(&sub, &m->subs[(size_t)r]);
        if (rc != NBE_OK) break;
        if (hipSetDevice(sub.device) != hipSuccess || hipEventCreateWithFlags(&m->advanced[(size_t)r], hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&m->pulled[(size_t)r], hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&m->vel_done[(size_t)r], hipEventDisableTiming) != hipSuccess ||
            hipStreamCreateWithFlags(&m->side[(size_t)r], hipStreamNonBlocking) != hipSuccess) {
            nbe_set_error("hipEventCreate / hipStreamCreate failed on device %d", sub.device);
            rc = NBE_E_DEVICE;
        }
    }
    // direct copies between distinct GPUs (without peer access hipMemcpyPeerAsync stages through the host, which
    // is still correct); "already enabled" is not an error
    for (int r = 0; r < G && rc == NBE_OK; r++)
        for (int s = 0; s < G; s++) {
            if (m->devs[(size_t)r] == m->devs[(size_t)s]) continue;
            int can = 0;
            if (hipDeviceCanAccessPeer(&can, m->devs[(size_t)r], m->devs[(size_t)s]) != hipSuccess || !can) continue;
            (void)hipSetDevice(m->devs[(size_t)r]);
            const hipError_t pe = hipDeviceEnablePeerAccess(m->devs[(size_t)s], 0);
            if (pe != hipSuccess) (void)hipGetLastError();      // e.g. hipErrorPeerAccessAlreadyEnabled
        }
    if (rc != NBE_OK) {
        char msg[512];
        snprintf(msg, sizeof(msg), "%s", nbe_last_error());
        multi_destroy(e);
        nbe_set_error("%s", msg);
        return rc;
    }
    *out = e;
    return NBE_OK;
}

int multi_upload(nbe_engine* e, size_t n, const double* pos, const double* vel, const double* mass, const uint64_t* id)
{
    MultiState* m = e->multi;
    const size_t G = m->subs.size();
    ranges(n, G, &m->first, &m->count);
    e->n = 0;                                   // until every engine holds the new bodies
    for (size_t r = 0; r < G; r++) {
        NBE_TRY(nbe_upload(m->subs[r], n, pos, vel, mass, id));
        NBE_TRY(nbe_set_shard(m->subs[r], m->first[r], m->count[r]));
    }
    e->n = n;
    return NBE_OK;
}

int multi_set_radius(nbe_engine* e, const double* radius)
{
    for (nbe_engine* s : e->multi->subs) NBE_TRY(nbe_set_radius(s, radius));
    return NBE_OK;
}

int multi_sync(nbe_engine* e)
{
    for (nbe_engine* s : e->multi->subs) NBE_TRY(nbe_sync(s));
    return NBE_OK;
}

int multi_step(nbe_engine* e, int nsteps)
{
    MultiState* m = e->multi;
    const size_t G = m->subs.size();
    if (e->n == 0) return NBE_OK;
    for (int step = 0; step < nsteps; step++) {
        // sort + build on every GPU (replicated, identical results); also sends the frame of this step from GPU 0
        for (size_t r = 0; r < G; r++) NBE_TRY(nbe_shard_prepare(m->subs[r]));
        for (size_t r = 0; r < G; r++) {
            nbe_engine* sub = m->subs[r];
            NBE_HIP(hipSetDevice(m->devs[r]));
            // the buffers this advance writes are the ones the others copied from in the previous step
            for (size_t s = 0; s < G; s++)
                if (s != r) NBE_HIP(hipStreamWaitEvent(sub->stream, m->pulled[s], 0));
            NBE_TRY(nbe_shard_advance(sub));
            NBE_HIP(hipEventRecord(m->advanced[r], sub->stream));
        }
        // Only the positions gate the next step (keys and radix passes read nothing else).  In tree mode every GPU
        // therefore pulls the positions on its own stream, commits, starts the next step's key sort, and pulls the
        // velocities on a side stream meanwhile; its own stream takes the velocities' arrival in before anything else.
        // With collisions on, the advance left accelerations only: every GPU pulls the others' ranges of those, and
        // the commit merges and updates all bodies on each GPU alike.
        const bool collide = e->cfg.collisions != 0;
        const bool overlap = e->cfg.mode == NBE_MODE_TREE && !collide;
        for (size_t s = 0; s < G; s++) {
            nbe_engine* dst = m->subs[s];
            NBE_HIP(hipSetDevice(m->devs[s]));
            nbe_device_state_t ds;
            NBE_TRY(nbe_device_state(dst, &ds));
            hipStream_t vstream = overlap ? m->side[s] : dst->stream;
            if (overlap) NBE_HIP(hipStreamWaitEvent(vstream, m->advanced[s], 0));     // behind this GPU's own use of the buffer
            for (size_t k = 1; k < G; k++) {
                const size_t r = (s + k) % G;                    // every GPU starts with a different source
                if (m->count[r] == 0) continue;
                nbe_device_state_t ss;
                NBE_TRY(nbe_device_state(m->subs[r], &ss));
                const size_t off = (size_t)m->first[r] * ds.elem_bytes, bytes = (size_t)m->count[r] * ds.elem_bytes;
                NBE_HIP(hipStreamWaitEvent(dst->stream, m->advanced[r], 0));
                if (collide) {
                    NBE_HIP(hipMemcpyPeerAsync((char*)ds.acc + off, m->devs[s], (const char*)ss.acc + off, m->devs[r], bytes, dst->stream));
                    continue;
                }
                NBE_HIP(hipMemcpyPeerAsync((char*)ds.pos_mass_next + off, m->devs[s], (const char*)ss.pos_mass_next + off, m->devs[r],
                                           bytes, dst->stream));
                if (overlap) NBE_HIP(hipStreamWaitEvent(vstream, m->advanced[r], 0));
                NBE_HIP(hipMemcpyPeerAsync((char*)ds.vel_next + off, m->devs[s], (const char*)ss.vel_next + off, m->devs[r], bytes,
                                           vstream));
            }
            if (overlap) NBE_HIP(hipEventRecord(m->vel_done[s], vstream));
        }
        // (the commit waits for the pair count of its GPU: queue every GPU's counting search first)
        if (collide)
            for (size_t r = 0; r < G; r++) {
                NBE_HIP(hipSetDevice(m->devs[r]));
                NBE_TRY(collide_begin(m->subs[r]));
            }
        for (size_t r = 0; r < G; r++) {
            nbe_engine* sub = m->subs[r];
            NBE_TRY(nbe_shard_commit(sub));
            NBE_HIP(hipSetDevice(m->devs[r]));
            if (overlap) {
                NBE_TRY(nbe_shard_sort(sub));                     // keys of the new positions, beside the velocity copies
                NBE_HIP(hipStreamWaitEvent(sub->stream, m->vel_done[r], 0));
            }
            NBE_HIP(hipEventRecord(m->pulled[r], sub->stream));   // this GPU no longer reads the others' buffers
        }
    }
    return NBE_OK;
}
