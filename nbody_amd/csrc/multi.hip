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
#include <cstdlib>

#include <algorithm>
#include <cstring>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

// One host thread per GPU for the distributed tree: every nbe_dist_* call ends by waiting for its GPU, so a single thread
// would run the GPUs one after the other.  run(f) calls f(r) on thread r for every r and returns the first failure; the
// failing thread's message is carried over to the caller's (messages are per thread).
struct Workers {
    std::vector<std::thread> threads;
    std::mutex mu;
    std::condition_variable go, done;
    std::function<int(int)> job;
    unsigned long long generation = 0;
    int pending = 0;
    bool stop = false;
    int first_rc = NBE_OK;
    std::string first_msg;

    void start(int n)
    {
        for (int r = 0; r < n; r++)
            threads.emplace_back([this, r]() {
                unsigned long long seen = 0;
                for (;;) {
                    std::function<int(int)> f;
                    {
                        std::unique_lock<std::mutex> lk(mu);
                        go.wait(lk, [&] { return stop || generation != seen; });
                        if (stop) return;
                        seen = generation;
                        f = job;
                    }
                    // a job allocates host vectors: an exception must not leave the thread (std::terminate)
                    int rc;
                    try {
                        rc = f(r);
                    } catch (const std::bad_alloc&) {
                        nbe_set_error("out of host memory in the worker of GPU %d", r);
                        rc = NBE_E_NOMEM;
                    } catch (const std::exception& ex) {
                        nbe_set_error("worker of GPU %d: %s", r, ex.what());
                        rc = NBE_E_INVALID;
                    }
                    {
                        std::lock_guard<std::mutex> lk(mu);
                        if (rc != NBE_OK && first_rc == NBE_OK) { first_rc = rc; first_msg = nbe_last_error(); }
                        if (--pending == 0) done.notify_all();
                    }
                }
            });
    }
    int run(std::function<int(int)> f)
    {
        std::unique_lock<std::mutex> lk(mu);
        job = std::move(f);
        pending = (int)threads.size();
        first_rc = NBE_OK;
        generation++;
        go.notify_all();
        done.wait(lk, [&] { return pending == 0; });
        if (first_rc != NBE_OK) nbe_set_error("%s", first_msg.c_str());
        return first_rc;
    }
    ~Workers()
    {
        {
            std::lock_guard<std::mutex> lk(mu);
            stop = true;
        }
        go.notify_all();
        for (std::thread& t : threads) t.join();
    }
};

// State of the distributed tree behind one handle (cfg.distributed): what disttree.py keeps per step, for all GPUs.
// Frames of a distributed run (main.go:90-101): no GPU holds all bodies, so every GPU sends its own, in rank order, into
// one pinned buffer; a delivery thread hands the whole frame to the host's callback.
struct DistFrameSlot {
    float* xyzm = nullptr;             // pinned, n_total x 4 floats
    uint32_t* orig = nullptr;          // pinned, n_total upload indices
    float* radius = nullptr;           // pinned, n_total radii (collisions on)
    std::vector<hipEvent_t> done;      // per GPU: its part has landed
    size_t n = 0;
    int frame = 0;
    bool busy = false;
};

struct DistFrames {
    bool active = false;
    int every = 1, first_frame = 0;
    uint64_t step_base = 0, last_k = 0;
    bool have_last = false;
    nbe_frame_cb cb = nullptr;
    void* user = nullptr;
    // per GPU: a snapshot of its bodies taken on its compute stream (the step that follows parks and overwrites them),
    // sent to the host on a stream of its own while the step computes
    std::vector<hipStream_t> side;
    std::vector<float4*> snap_xyzm;
    std::vector<uint32_t*> snap_orig;
    std::vector<float*> snap_radius;                 // collisions on
    std::vector<hipEvent_t> snapped, sent;
    std::vector<char> sending;
    std::vector<DistFrameSlot> slots;
    std::deque<int> queue;
    std::mutex mu;
    std::condition_variable cv_work, cv_free;
    std::thread worker;
    bool stop = false;
    std::vector<uint64_t> ids_scratch;
};

struct DistMulti {
    Workers pool;
    DistFrames frames;
    size_t n_total = 0;
    std::vector<uint64_t> ids;                       // upload order
    std::vector<size_t> capacity;
    std::vector<nbe_dist_buffers_t> bufs;
    std::vector<std::vector<uint32_t>> send;         // [from][to] migrating bodies
    std::vector<std::vector<nbe_fragment_t>> frags;
    std::vector<std::vector<double>> boxes;          // [rank][NBE_DIST_MAX_BOXES * 6]
    std::vector<int> nboxes, nfrags;
    std::vector<uint64_t> live;
    std::vector<std::vector<int32_t>> sizes;         // [from][to * nfrags[from] + f]
    std::vector<std::vector<uint64_t>> offsets;      // [from][to]
    bool prepared = false, want_rebalance = false;
    int rebalances = 0;
};

struct MultiState {
    std::vector<nbe_engine*> subs;
    std::vector<int> devs;
    std::vector<uint64_t> first, count;      // range of device-order bodies each engine advances
    std::vector<hipEvent_t> advanced;        // engine r: its own range of *_next is written (recorded on its stream)
    std::vector<hipEvent_t> pulled;          // engine r: it has finished copying the other ranges (recorded on its stream)
    std::vector<hipStream_t> side;           // engine r: stream of its velocity copies (tree mode: they run beside the key sort)
    std::vector<hipEvent_t> vel_done;        // engine r: its velocity copies have landed (recorded on side[r])
    DistMulti* dist = nullptr;               // cfg.distributed: the bodies are not replicated
    bool poisoned = false;                   // a step failed part-way: the GPUs may disagree about the state until the next upload
};

nbe_engine* multi_first(nbe_engine* e) { return e->multi->subs[0]; }

static void ranges(size_t n, size_t parts, std::vector<uint64_t>* first, std::vector<uint64_t>* count)
{
    // Ranges start at multiples of 256 bodies, so that a wavefront of the tree walk holds the same bodies as in a
    // one-GPU run: the per-wave criterion (NBE_WALK_PER_WAVE) decides for aligned groups of 128 bodies.  (The per-body
    // walk gives a body the same bits whichever bodies share its wavefront, in every precision.)
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
        if (m->dist) (void)multi_dist_frames_end(e);
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
        delete m->dist;
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
    if (cfg->distributed && (cfg->mode != NBE_MODE_TREE || cfg->reference_quirks || cfg->walk_criterion != NBE_WALK_PER_BODY)) {
        nbe_set_error("distributed = 1 needs tree mode, the per-body criterion and no reference_quirks");
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
        sub.distributed = 0;
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
    if (cfg->distributed) {
        m->dist = new (std::nothrow) DistMulti();
        if (!m->dist) { multi_destroy(e); nbe_set_error("out of host memory"); return NBE_E_NOMEM; }
        m->dist->pool.start(G);
    }
    *out = e;
    return NBE_OK;
}

bool multi_is_distributed(const nbe_engine* e) { return e->multi && e->multi->dist; }

// ---- the distributed tree on the engine's GPUs --------------------------------------------------------------------
// Splitters from every GPU's evenly spaced sorted keys: a sample of GPU r stands for live[r] / samples bodies; the k-th
// splitter is the key where the running weight passes k/G of all bodies (a splitter is a key: equal keys stay together).
static std::vector<uint64_t> choose_splitters(const std::vector<std::vector<uint64_t>>& samples, const std::vector<size_t>& live)
{
    const size_t G = samples.size();
    std::vector<std::pair<uint64_t, double>> all;
    for (size_t r = 0; r < G; r++)
        for (uint64_t k : samples[r]) all.push_back({k, (double)live[r] / (double)samples[r].size()});
    std::stable_sort(all.begin(), all.end(), [](const std::pair<uint64_t, double>& a, const std::pair<uint64_t, double>& b) { return a.first < b.first; });
    std::vector<uint64_t> split(G + 1, 0);
    split[G] = 1ull << 63;
    if (all.empty()) {
        for (size_t k = 1; k < G; k++) split[k] = split[G];
        return split;
    }
    double total = 0;
    for (const auto& a : all) total += a.second;
    double run = 0;
    size_t j = 0;
    for (size_t k = 1; k < G; k++) {
        const double want = total * (double)k / (double)G;
        while (j + 1 < all.size() && run + all[j].second < want) { run += all[j].second; j++; }
        split[k] = std::max(all[j].first, split[k - 1]);
    }
    return split;
}

static int dist_distribute(nbe_engine* e)
{
    MultiState* m = e->multi;
    DistMulti* d = m->dist;
    const size_t G = m->subs.size();
    std::vector<std::vector<uint64_t>> samples(G);
    std::vector<size_t> live(G, 0);
    NBE_TRY(d->pool.run([&](int r) -> int {
        samples[(size_t)r].assign(4096, 0);
        size_t got = 0;
        NBE_TRY(nbe_dist_sample_keys(m->subs[(size_t)r], 4096, samples[(size_t)r].data(), &got, &live[(size_t)r]));
        samples[(size_t)r].resize(got);
        return NBE_OK;
    }));
    const std::vector<uint64_t> split = choose_splitters(samples, live);
    return d->pool.run([&](int r) -> int { return nbe_dist_set_splitters(m->subs[(size_t)r], split.data()); });
}

static int dist_upload(nbe_engine* e, size_t n, const double* pos, const double* vel, const double* mass, const uint64_t* id)
{
    MultiState* m = e->multi;
    DistMulti* d = m->dist;
    const size_t G = m->subs.size();
    const size_t per = (n + G - 1) / G;
    NBE_TRY(multi_dist_frames_end(e));               // (as an upload ends the frame stream of one engine)
    e->n = 0;
    d->prepared = false; d->want_rebalance = false;
    d->n_total = n;
    d->ids.resize(n);
    for (size_t i = 0; i < n; i++) d->ids[i] = id ? id[i] : (uint64_t)i;
    // room for a quarter more than an even share; NBODY_DIST_HEADROOM = h (1.25 .. number of GPUs) gives h even shares
    // instead, for scenes that cannot be shared out evenly (with h = number of GPUs any distribution fits)
    double headroom = 1.25;
    if (const char* env = getenv("NBODY_DIST_HEADROOM")) headroom = std::min(std::max(atof(env), 1.25), (double)G);
    d->capacity.assign(G, (size_t)((double)per * headroom) + 4096);
    d->bufs.assign(G, nbe_dist_buffers_t{});
    double total_mass = 0.0;                         // fixes the quantum of the exact cell sums, alike on every GPU
    for (size_t i = 0; i < n; i++) total_mass += mass[i];
    NBE_TRY(d->pool.run([&](int r) -> int {
        const size_t a = std::min(n, (size_t)r * per), z = std::min(n, ((size_t)r + 1) * per);
        NBE_TRY(nbe_dist_upload(m->subs[(size_t)r], r, (int)G, d->capacity[(size_t)r], z - a, pos + 3 * a, vel + 3 * a, mass + a, a));
        NBE_TRY(nbe_dist_set_total_mass(m->subs[(size_t)r], total_mass));
        return nbe_dist_buffers(m->subs[(size_t)r], &d->bufs[(size_t)r]);
    }));
    NBE_TRY(dist_distribute(e));
    e->n = n;
    return NBE_OK;
}

// Everything of a step but the force pass (what DistributedTree._prepare does in nbody_amd/disttree.py).
static int dist_prepare(nbe_engine* e)
{
    MultiState* m = e->multi;
    DistMulti* d = m->dist;
    const size_t G = m->subs.size();
    if (d->want_rebalance) {
        NBE_TRY(dist_distribute(e));
        d->rebalances++;
        d->want_rebalance = false;
    }
    // bodies that crossed a splitter go to their new owner
    for (int attempt = 0;; attempt++) {
        d->send.assign(G, std::vector<uint32_t>(G, 0));
        std::vector<uint64_t> room(G, 0);
        NBE_TRY(d->pool.run([&](int r) -> int {
            NBE_TRY(nbe_dist_classify(m->subs[(size_t)r], d->send[(size_t)r].data()));
            return nbe_dist_room(m->subs[(size_t)r], &room[(size_t)r]);
        }));
        bool fits = true;
        for (size_t q = 0; q < G; q++) {
            uint64_t bound = 0;
            for (size_t r = 0; r < G; r++) bound += d->send[r][q];
            if (bound > room[q]) fits = false;
        }
        if (fits) break;
        // More bodies are bound for some GPU than it has room for: the distribution over key space changed more in one
        // step than the spare slots allow.  The leavers go back to their slots, new splitters are chosen from the bodies
        // as they are now, and the classification is done again.
        if (attempt == 1) {
            nbe_set_error("the GPUs' capacities do not hold the bodies even with new splitters (NBODY_DIST_HEADROOM, read at nbe_upload, gives every GPU more room)%s",
                          e->cfg.collisions ? " (with collisions on a splitter sits on the border of a cell at least four radius caps wide: "
                                              "the bodies may span too few such cells to be shared out evenly)" : "");
            return NBE_E_NOMEM;
        }
        NBE_TRY(d->pool.run([&](int r) -> int { return nbe_dist_unclassify(m->subs[(size_t)r]); }));
        NBE_TRY(dist_distribute(e));
        d->rebalances++;
    }
    d->frags.assign(G, std::vector<nbe_fragment_t>(NBE_DIST_MAX_FRAGMENTS));
    d->boxes.assign(G, std::vector<double>(NBE_DIST_MAX_BOXES * 6, 0.0));
    d->nboxes.assign(G, 0); d->nfrags.assign(G, 0); d->live.assign(G, 0);
    NBE_TRY(d->pool.run([&](int qi) -> int {
        const size_t q = (size_t)qi;
        nbe_engine* sub = m->subs[q];
        NBE_HIP(hipSetDevice(m->devs[q]));
        const size_t rec = d->bufs[q].record_bytes;
        size_t at = 0;
        for (size_t r = 0; r < G; r++) {
            const size_t k = d->send[r][q];
            if (!k) continue;
            size_t off = 0;
            for (size_t t = 0; t < q; t++) off += d->send[r][t];
            if (at + k > d->bufs[q].slots) { nbe_set_error("GPU %zu: %zu bodies arrive, its inbox holds %llu", q, at + k, (unsigned long long)d->bufs[q].slots); return NBE_E_NOMEM; }
            NBE_HIP(hipMemcpyPeerAsync((char*)d->bufs[q].inbox + at * rec, m->devs[q], (const char*)d->bufs[r].outbox + off * rec, m->devs[r],
                                       k * rec, sub->stream));
            at += k;
        }
        NBE_TRY(nbe_dist_receive(sub, at));
        return nbe_dist_build(sub, d->frags[q].data(), NBE_DIST_MAX_FRAGMENTS, &d->nfrags[q], d->boxes[q].data(), &d->nboxes[q], &d->live[q]);
    }));
    // every GPU needs every GPU's fragment roots and boxes
    std::vector<nbe_fragment_t> all;
    std::vector<double> boxes(G * NBE_DIST_MAX_BOXES * 6, 0.0);
    uint64_t total_live = 0, most = 0;
    double fill = 0;
    for (size_t r = 0; r < G; r++) {
        all.insert(all.end(), d->frags[r].begin(), d->frags[r].begin() + d->nfrags[r]);
        std::copy(d->boxes[r].begin(), d->boxes[r].end(), boxes.begin() + (long)(r * NBE_DIST_MAX_BOXES * 6));
        total_live += d->live[r];
        most = std::max(most, d->live[r]);
        fill = std::max(fill, (double)d->live[r] / (double)d->capacity[r]);
    }
    // new splitters before the next step when a GPU holds a tenth more than its share or has filled nine tenths of its slots
    d->want_rebalance = total_live > 0 && ((double)most > 1.10 * (double)total_live / (double)G + 64.0 || fill > 0.9);
    d->sizes.assign(G, std::vector<int32_t>());
    d->offsets.assign(G, std::vector<uint64_t>(G + 1, 0));
    NBE_TRY(d->pool.run([&](int r) -> int {
        d->sizes[(size_t)r].assign(G * (size_t)std::max(d->nfrags[(size_t)r], 1), 0);
        return nbe_dist_export(m->subs[(size_t)r], all.data(), d->nfrags.data(), boxes.data(), d->nboxes.data(), d->sizes[(size_t)r].data(),
                               d->offsets[(size_t)r].data());
    }));
    NBE_TRY(d->pool.run([&](int qi) -> int {
        const size_t q = (size_t)qi;
        nbe_engine* sub = m->subs[q];
        NBE_HIP(hipSetDevice(m->devs[q]));
        const size_t nb = d->bufs[q].node_bytes;
        std::vector<uint64_t> import_offsets(G, 0);
        std::vector<int32_t> import_sizes;
        size_t at = 0;
        for (size_t r = 0; r < G; r++) {
            import_offsets[r] = at;
            const size_t nf = (size_t)d->nfrags[r];
            for (size_t f = 0; f < nf; f++) import_sizes.push_back(r == q ? 0 : d->sizes[r][q * nf + f]);
            if (r == q) continue;
            const size_t k = (size_t)(d->offsets[r][q + 1] - d->offsets[r][q]);
            if (!k) continue;
            if (at + k > d->bufs[q].import_cap) { nbe_set_error("GPU %zu: %zu nodes arrive, its import buffer holds %llu", q, at + k, (unsigned long long)d->bufs[q].import_cap); return NBE_E_NOMEM; }
            NBE_HIP(hipMemcpyPeerAsync((char*)d->bufs[q].import_nodes + at * nb, m->devs[q],
                                       (const char*)d->bufs[r].export_nodes + (size_t)d->offsets[r][q] * nb, m->devs[r], k * nb, sub->stream));
            at += k;
        }
        if (import_sizes.empty()) import_sizes.push_back(0);
        return nbe_dist_assemble(sub, import_sizes.data(), import_offsets.data());
    }));
    d->prepared = true;
    return NBE_OK;
}

// ---- frames of a distributed run ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void frame_narrow_kernel(const double4* __restrict__ posm, float4* __restrict__ out, int n)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const double4 p = posm[i];
    out[i] = make_float4((float)p.x, (float)p.y, (float)p.z, (float)p.w);
}

__global__ __launch_bounds__(256) void frame_narrow_radius_kernel(const double* __restrict__ radius, float* __restrict__ out, int n)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = (float)radius[i];
}

static void dist_frames_deliver(nbe_engine* e)
{
    DistMulti* d = e->multi->dist;
    DistFrames& fs = d->frames;
    for (;;) {
        int slot;
        {
            std::unique_lock<std::mutex> lk(fs.mu);
            fs.cv_work.wait(lk, [&] { return fs.stop || !fs.queue.empty(); });
            if (fs.queue.empty()) return;            // stop requested and nothing left to deliver
            slot = fs.queue.front();
            fs.queue.pop_front();
        }
        DistFrameSlot& s = fs.slots[(size_t)slot];
        for (hipEvent_t ev : s.done) (void)hipEventSynchronize(ev);
        if (e->cfg.collisions) {
            // a body absorbed in the previous step still holds its slot until its GPU's next build: massless and parked
            // outside the bound (collide_update_kernel); it is no body of the frame (main.go:91-95 skips the nil entries)
            const float parked = (float)(e->cfg.bound_center[0] + 1.5 * e->cfg.bound_width);
            size_t keep = 0;
            for (size_t k = 0; k < s.n; k++) {
                if (s.xyzm[4 * k + 3] == 0.0f && s.xyzm[4 * k] == parked) continue;
                if (keep != k) { memcpy(s.xyzm + 4 * keep, s.xyzm + 4 * k, 4 * sizeof(float)); s.orig[keep] = s.orig[k]; s.radius[keep] = s.radius[k]; }
                keep++;
            }
            s.n = keep;
        }
        fs.ids_scratch.resize(s.n);
        for (size_t k = 0; k < s.n; k++) fs.ids_scratch[k] = s.orig[k] < d->ids.size() ? d->ids[s.orig[k]] : (uint64_t)s.orig[k];
        fs.cb(fs.user, s.frame, s.n, s.xyzm, e->cfg.collisions ? s.radius : nullptr, fs.ids_scratch.data());
        {
            std::lock_guard<std::mutex> lk(fs.mu);
            s.busy = false;
        }
        fs.cv_free.notify_all();
    }
}

static void dist_frames_release(nbe_engine* e)
{
    MultiState* m = e->multi;
    DistFrames& fs = m->dist->frames;
    for (size_t g = 0; g < m->subs.size(); g++) {
        (void)hipSetDevice(m->devs[g]);
        if (g < fs.side.size() && fs.side[g]) { (void)hipStreamSynchronize(fs.side[g]); (void)hipStreamDestroy(fs.side[g]); }
        if (g < fs.snap_xyzm.size() && fs.snap_xyzm[g]) (void)hipFree(fs.snap_xyzm[g]);
        if (g < fs.snap_orig.size() && fs.snap_orig[g]) (void)hipFree(fs.snap_orig[g]);
        if (g < fs.snap_radius.size() && fs.snap_radius[g]) (void)hipFree(fs.snap_radius[g]);
        if (g < fs.snapped.size() && fs.snapped[g]) (void)hipEventDestroy(fs.snapped[g]);
        if (g < fs.sent.size() && fs.sent[g]) (void)hipEventDestroy(fs.sent[g]);
        for (DistFrameSlot& s : fs.slots)
            if (g < s.done.size() && s.done[g]) (void)hipEventDestroy(s.done[g]);
    }
    for (DistFrameSlot& s : fs.slots) {
        if (s.xyzm) (void)hipHostFree(s.xyzm);
        if (s.orig) (void)hipHostFree(s.orig);
        if (s.radius) (void)hipHostFree(s.radius);
    }
    fs.slots.clear();
    fs.side.clear(); fs.snap_xyzm.clear(); fs.snap_orig.clear(); fs.snap_radius.clear(); fs.snapped.clear(); fs.sent.clear(); fs.sending.clear();
}

static int dist_frames_allocate(nbe_engine* e, int slots)
{
    MultiState* m = e->multi;
    DistMulti* d = m->dist;
    DistFrames& fs = d->frames;
    const size_t G = m->subs.size();
    const size_t total = d->n_total > 0 ? d->n_total : 1;
    fs.side.assign(G, nullptr); fs.snap_xyzm.assign(G, nullptr); fs.snap_orig.assign(G, nullptr); fs.snap_radius.assign(G, nullptr);
    fs.snapped.assign(G, nullptr); fs.sent.assign(G, nullptr); fs.sending.assign(G, 0);
    fs.slots.assign((size_t)slots, DistFrameSlot());
    for (DistFrameSlot& s : fs.slots) s.done.assign(G, nullptr);
    for (size_t g = 0; g < G; g++) {
        NBE_HIP(hipSetDevice(m->devs[g]));
        const size_t cap = g < d->capacity.size() && d->capacity[g] > 0 ? d->capacity[g] : 1;
        NBE_HIP(hipStreamCreateWithFlags(&fs.side[g], hipStreamNonBlocking));
        NBE_HIP(hipMalloc((void**)&fs.snap_xyzm[g], cap * sizeof(float4)));
        NBE_HIP(hipMalloc((void**)&fs.snap_orig[g], cap * sizeof(uint32_t)));
        if (e->cfg.collisions) NBE_HIP(hipMalloc((void**)&fs.snap_radius[g], cap * sizeof(float)));
        NBE_HIP(hipEventCreateWithFlags(&fs.snapped[g], hipEventDisableTiming));
        NBE_HIP(hipEventCreateWithFlags(&fs.sent[g], hipEventDisableTiming));
        for (DistFrameSlot& s : fs.slots) NBE_HIP(hipEventCreateWithFlags(&s.done[g], hipEventDisableTiming));
    }
    for (DistFrameSlot& s : fs.slots) {
        NBE_HIP(hipHostMalloc((void**)&s.xyzm, total * 4 * sizeof(float), hipHostMallocPortable));
        NBE_HIP(hipHostMalloc((void**)&s.orig, total * sizeof(uint32_t), hipHostMallocPortable));
        if (e->cfg.collisions) NBE_HIP(hipHostMalloc((void**)&s.radius, total * sizeof(float), hipHostMallocPortable));
    }
    return NBE_OK;
}

int multi_dist_frames_begin(nbe_engine* e, int every, int slots, int first_frame, nbe_frame_cb cb, void* user)
{
    MultiState* m = e->multi;
    DistFrames& fs = m->dist->frames;
    if (every < 1 || slots < 1 || slots > 1024) { nbe_set_error("every must be >= 1 and slots in 1..1024"); return NBE_E_INVALID; }
    if (fs.active) { nbe_set_error("frame streaming is already on"); return NBE_E_INVALID; }
    fs.every = every; fs.first_frame = first_frame; fs.cb = cb; fs.user = user;
    fs.step_base = m->subs[0]->stats.steps;
    fs.have_last = false; fs.stop = false;
    fs.queue.clear();
    int rc;
    try {
        rc = dist_frames_allocate(e, slots);
        if (rc == NBE_OK) {
            fs.active = true;
            fs.worker = std::thread(dist_frames_deliver, e);
        }
    } catch (const std::exception& ex) {
        nbe_set_error("frame streaming could not start: %s", ex.what());
        rc = NBE_E_NOMEM;
    }
    if (rc != NBE_OK) {
        fs.active = false;
        dist_frames_release(e);
    }
    return rc;
}

int multi_dist_frames_end(nbe_engine* e)
{
    DistFrames& fs = e->multi->dist->frames;
    if (!fs.active) return NBE_OK;
    {
        std::lock_guard<std::mutex> lk(fs.mu);
        fs.stop = true;
    }
    fs.cv_work.notify_all();
    if (fs.worker.joinable()) fs.worker.join();      // delivers everything still queued first
    dist_frames_release(e);
    fs.active = false;
    return NBE_OK;
}

// Called at the top of every step: the bodies as the step is about to read them.  Between a build and the next classify
// a GPU's bodies fill its first slots in key order, and the GPUs follow each other in key order.
static int dist_frames_before_step(nbe_engine* e)
{
    MultiState* m = e->multi;
    DistMulti* d = m->dist;
    DistFrames& fs = d->frames;
    if (!fs.active) return NBE_OK;
    const uint64_t k = m->subs[0]->stats.steps - fs.step_base;
    if (k % (uint64_t)fs.every != 0) return NBE_OK;
    if (fs.have_last && fs.last_k == k) return NBE_OK;
    const size_t G = m->subs.size();
    std::vector<size_t> live(G, 0);
    size_t total = 0;
    for (size_t g = 0; g < G; g++) { live[g] = dist_live_slots(m->subs[g]); total += live[g]; }
    if (d->n_total == 0) return NBE_OK;              // (a run whose bodies have all left the bound still sends its frames, empty)
    if (total > d->n_total) { nbe_set_error("the GPUs hold %zu bodies of %zu", total, d->n_total); return NBE_E_DEVICE; }
    fs.have_last = true; fs.last_k = k;
    int slot = -1;
    {
        std::unique_lock<std::mutex> lk(fs.mu);      // back-pressure: wait for a free buffer like a full channel
        fs.cv_free.wait(lk, [&] {
            for (size_t i = 0; i < fs.slots.size(); i++)
                if (!fs.slots[i].busy) { slot = (int)i; return true; }
            return false;
        });
        fs.slots[(size_t)slot].busy = true;
    }
    DistFrameSlot& s = fs.slots[(size_t)slot];
    s.frame = fs.first_frame + (int)(k / (uint64_t)fs.every);
    s.n = total;
    size_t at = 0;
    for (size_t g = 0; g < G; g++) {
        nbe_engine* sub = m->subs[g];
        const size_t n = live[g];
        NBE_HIP(hipSetDevice(m->devs[g]));
        if (fs.sending[g]) NBE_HIP(hipStreamWaitEvent(sub->stream, fs.sent[g], 0));      // the snapshot is still being sent
        if (n > 0) {
            if (sub->state64) {
                hipLaunchKernelGGL(frame_narrow_kernel, dim3(cdiv(n, 256)), dim3(256), 0, sub->stream, (const double4*)sub->posm[sub->cur],
                                   fs.snap_xyzm[g], (int)n);
                NBE_HIP(hipGetLastError());
            } else {
                NBE_HIP(hipMemcpyAsync(fs.snap_xyzm[g], sub->posm[sub->cur], n * sizeof(float4), hipMemcpyDeviceToDevice, sub->stream));
            }
            NBE_HIP(hipMemcpyAsync(fs.snap_orig[g], sub->orig[sub->orig_cur], n * sizeof(uint32_t), hipMemcpyDeviceToDevice, sub->stream));
            if (e->cfg.collisions) {
                hipLaunchKernelGGL(frame_narrow_radius_kernel, dim3(cdiv(n, 256)), dim3(256), 0, sub->stream, (const double*)sub->radius[sub->cur],
                                   fs.snap_radius[g], (int)n);
                NBE_HIP(hipGetLastError());
            }
        }
        NBE_HIP(hipEventRecord(fs.snapped[g], sub->stream));
        NBE_HIP(hipStreamWaitEvent(fs.side[g], fs.snapped[g], 0));
        if (n > 0) {
            NBE_HIP(hipMemcpyAsync(s.xyzm + 4 * at, fs.snap_xyzm[g], n * sizeof(float4), hipMemcpyDeviceToHost, fs.side[g]));
            NBE_HIP(hipMemcpyAsync(s.orig + at, fs.snap_orig[g], n * sizeof(uint32_t), hipMemcpyDeviceToHost, fs.side[g]));
            if (e->cfg.collisions) NBE_HIP(hipMemcpyAsync(s.radius + at, fs.snap_radius[g], n * sizeof(float), hipMemcpyDeviceToHost, fs.side[g]));
        }
        NBE_HIP(hipEventRecord(s.done[g], fs.side[g]));
        NBE_HIP(hipEventRecord(fs.sent[g], fs.side[g]));
        fs.sending[g] = 1;
        at += n;
    }
    {
        std::lock_guard<std::mutex> lk(fs.mu);
        fs.queue.push_back(slot);
    }
    fs.cv_work.notify_one();
    return NBE_OK;
}

static int dist_step(nbe_engine* e, int nsteps)
{
    MultiState* m = e->multi;
    DistMulti* d = m->dist;
    for (int s = 0; s < nsteps; s++) {
        NBE_TRY(dist_frames_before_step(e));
        if (!d->prepared) NBE_TRY(dist_prepare(e));
        NBE_TRY(d->pool.run([&](int r) -> int { return nbe_dist_advance(m->subs[(size_t)r]); }));
        d->prepared = false;
    }
    return NBE_OK;
}

int multi_dist_download(nbe_engine* e, double* pos, double* vel, double* mass, uint64_t* id, size_t* n_alive_out)
{
    MultiState* m = e->multi;
    DistMulti* d = m->dist;
    const size_t G = m->subs.size();
    if (n_alive_out) *n_alive_out = 0;
    if (d->capacity.size() != G) return NBE_OK;       // nothing uploaded yet
    std::vector<std::vector<uint64_t>> idx(G);
    std::vector<std::vector<double>> p(G), v(G), ms(G);
    std::vector<size_t> cnt(G, 0);
    NBE_TRY(d->pool.run([&](int r) -> int {
        const size_t cap = d->capacity[(size_t)r];
        idx[(size_t)r].resize(cap); p[(size_t)r].resize(3 * cap); v[(size_t)r].resize(3 * cap); ms[(size_t)r].resize(cap);
        return nbe_dist_download(m->subs[(size_t)r], idx[(size_t)r].data(), p[(size_t)r].data(), v[(size_t)r].data(), ms[(size_t)r].data(), &cnt[(size_t)r]);
    }));
    // survivors in upload order, as one engine returns them
    std::vector<std::pair<uint64_t, std::pair<uint32_t, uint32_t>>> where;
    for (size_t r = 0; r < G; r++)
        for (size_t k = 0; k < cnt[r]; k++) where.push_back({idx[r][k], {(uint32_t)r, (uint32_t)k}});
    std::sort(where.begin(), where.end());
    for (size_t o = 0; o < where.size(); o++) {
        const size_t r = where[o].second.first, k = where[o].second.second;
        for (int c = 0; c < 3; c++) {
            if (pos) pos[3 * o + (size_t)c] = p[r][3 * k + (size_t)c];
            if (vel) vel[3 * o + (size_t)c] = v[r][3 * k + (size_t)c];
        }
        if (mass) mass[o] = ms[r][k];
        if (id) id[o] = where[o].first < d->ids.size() ? d->ids[where[o].first] : where[o].first;
    }
    if (n_alive_out) *n_alive_out = where.size();
    return NBE_OK;
}

// nbe_accel on the distributed tree: every GPU's accelerations of its own bodies, put in upload order; bodies that are
// gone (outside the bound, absorbed) get zeros and alive = 0.
int multi_dist_accel(nbe_engine* e, double* acc_xyz, uint8_t* alive_out)
{
    MultiState* m = e->multi;
    DistMulti* d = m->dist;
    const size_t G = m->subs.size(), n = d->n_total;
    if (n == 0 || d->capacity.size() != G) return NBE_OK;
    if (!acc_xyz) { nbe_set_error("acc_xyz is NULL"); return NBE_E_INVALID; }
    if (!d->prepared) NBE_TRY(dist_prepare(e));
    std::vector<std::vector<uint64_t>> idx(G);
    std::vector<std::vector<double>> acc(G);
    std::vector<size_t> cnt(G, 0);
    NBE_TRY(d->pool.run([&](int r) -> int {
        const size_t cap = d->capacity[(size_t)r];
        idx[(size_t)r].resize(cap); acc[(size_t)r].resize(3 * cap);
        return nbe_dist_accel(m->subs[(size_t)r], idx[(size_t)r].data(), acc[(size_t)r].data(), &cnt[(size_t)r]);
    }));
    for (size_t u = 0; u < n; u++) {
        acc_xyz[3 * u] = acc_xyz[3 * u + 1] = acc_xyz[3 * u + 2] = 0.0;
        if (alive_out) alive_out[u] = 0;
    }
    for (size_t r = 0; r < G; r++)
        for (size_t k = 0; k < cnt[r]; k++) {
            const uint64_t u = idx[r][k];
            if (u >= n) continue;
            for (int c = 0; c < 3; c++) acc_xyz[3 * u + (size_t)c] = acc[r][3 * k + (size_t)c];
            if (alive_out) alive_out[u] = 1;
        }
    return NBE_OK;
}

// Radii of the survivors in upload order (the order of multi_dist_download).
int multi_dist_get_radius(nbe_engine* e, double* radius, size_t* n_alive_out)
{
    MultiState* m = e->multi;
    DistMulti* d = m->dist;
    const size_t G = m->subs.size();
    if (!e->cfg.collisions) { nbe_set_error("nbe_get_radius needs an engine created with collisions on"); return NBE_E_INVALID; }
    if (n_alive_out) *n_alive_out = 0;
    if (e->n == 0 || d->capacity.size() != G) return NBE_OK;
    std::vector<std::vector<uint64_t>> idx(G);
    std::vector<std::vector<double>> rad(G);
    std::vector<size_t> cnt(G, 0);
    NBE_TRY(d->pool.run([&](int r) -> int {
        const size_t cap = d->capacity[(size_t)r];
        idx[(size_t)r].resize(cap); rad[(size_t)r].resize(cap);
        return nbe_dist_get_radius(m->subs[(size_t)r], idx[(size_t)r].data(), rad[(size_t)r].data(), &cnt[(size_t)r]);
    }));
    std::vector<std::pair<uint64_t, double>> where;
    for (size_t r = 0; r < G; r++)
        for (size_t k = 0; k < cnt[r]; k++) where.push_back({idx[r][k], rad[r][k]});
    std::sort(where.begin(), where.end());
    if (radius)
        for (size_t o = 0; o < where.size(); o++) radius[o] = where[o].second;
    if (n_alive_out) *n_alive_out = where.size();
    return NBE_OK;
}

int multi_dist_energy(nbe_engine* e, double* ke, double* pe, double mom[3])
{
    MultiState* m = e->multi;
    DistMulti* d = m->dist;
    const size_t G = m->subs.size();
    if (e->n == 0) { if (ke) *ke = 0; if (pe) *pe = 0; if (mom) mom[0] = mom[1] = mom[2] = 0; return NBE_OK; }
    if (!d->prepared) NBE_TRY(dist_prepare(e));
    std::vector<double> part(G * 5, 0.0);
    NBE_TRY(d->pool.run([&](int r) -> int { return nbe_dist_energy(m->subs[(size_t)r], &part[5 * (size_t)r], &part[5 * (size_t)r + 1], &part[5 * (size_t)r + 2]); }));
    double tot[5] = {0, 0, 0, 0, 0};
    for (size_t r = 0; r < G; r++)
        for (int k = 0; k < 5; k++) tot[k] += part[5 * r + (size_t)k];
    if (ke) *ke = tot[0];
    if (pe) *pe = tot[1];
    if (mom) { mom[0] = tot[2]; mom[1] = tot[3]; mom[2] = tot[4]; }
    return NBE_OK;
}

int multi_dist_stats(nbe_engine* e, nbe_stats_t* out)
{
    MultiState* m = e->multi;
    memset(out, 0, sizeof(*out));
    out->n = m->dist->n_total;
    for (nbe_engine* s : m->subs) {
        if (e->cfg.collisions) {
            nbe_stats_t st;
            NBE_TRY(nbe_stats(s, &st));           // (reads the GPU's counts: bodies absorbed since its last build)
            out->merges += st.merges;
        }
        out->n_alive += s->stats.n_alive;
        out->tree_nodes += s->stats.tree_nodes;
        out->tree_cells += s->stats.tree_cells;
        out->collision_pairs += s->stats.collision_pairs;
    }
    out->steps = m->subs[0]->stats.steps;
    return NBE_OK;
}

// Waits for everything any GPU of the engine has queued: the engines' own streams and the side streams of the velocity
// copies.  An upload frees and reallocates every engine's buffers, and hipFree waits for the owning GPU's streams only,
// while the other GPUs' peer copies read those buffers from streams of their own.
static int multi_quiesce(nbe_engine* e)
{
    MultiState* m = e->multi;
    for (size_t r = 0; r < m->subs.size(); r++) {
        if (m->subs[r]) NBE_TRY(nbe_sync(m->subs[r]));
        if (r < m->side.size() && m->side[r]) {
            NBE_HIP(hipSetDevice(m->devs[r]));
            NBE_HIP(hipStreamSynchronize(m->side[r]));
        }
    }
    return NBE_OK;
}

int multi_upload(nbe_engine* e, size_t n, const double* pos, const double* vel, const double* mass, const uint64_t* id)
{
    MultiState* m = e->multi;
    NBE_TRY(multi_quiesce(e));
    m->poisoned = false;
    if (m->dist) return dist_upload(e, n, pos, vel, mass, id);
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
    MultiState* m = e->multi;
    if (m->dist) {
        // every GPU takes the radii of the bodies it was handed at the upload, and all agree on the largest radius a body
        // may grow to (sixteen times the largest given, at least 32 m): the splitters are then chosen anew, aligned for it
        DistMulti* d = m->dist;
        const size_t G = m->subs.size(), n = d->n_total, per = (n + G - 1) / G;
        if (!e->cfg.collisions) { nbe_set_error("nbe_set_radius needs an engine created with collisions on"); return NBE_E_INVALID; }
        if (n > 0 && !radius) { nbe_set_error("radius is NULL"); return NBE_E_INVALID; }
        if (d->capacity.size() != G) return NBE_OK;      // nothing uploaded yet
        if (m->subs[0]->stats.steps != 0 || d->prepared) { nbe_set_error("with distributed = 1, nbe_set_radius comes right after nbe_upload"); return NBE_E_INVALID; }
        double cap = 2.0;
        for (size_t i = 0; i < n; i++) {
            if (!(radius[i] >= 0.0)) { nbe_set_error("radius[%zu] is negative or NaN", i); return NBE_E_INVALID; }
            cap = std::max(cap, radius[i]);
        }
        cap *= 16.0;
        NBE_TRY(d->pool.run([&](int r) -> int {
            const size_t a = std::min(n, (size_t)r * per);
            return nbe_dist_set_radius(m->subs[(size_t)r], radius + a, cap);
        }));
        return dist_distribute(e);
    }
    for (nbe_engine* s : m->subs) NBE_TRY(nbe_set_radius(s, radius));
    return NBE_OK;
}

int multi_sync(nbe_engine* e)
{
    for (nbe_engine* s : e->multi->subs) NBE_TRY(nbe_sync(s));
    return NBE_OK;
}

static int multi_step_replicated(nbe_engine* e, int nsteps);

int multi_step(nbe_engine* e, int nsteps)
{
    MultiState* m = e->multi;
    if (m->poisoned) {
        nbe_set_error("an earlier nbe_step failed part-way and the GPUs may hold different states: upload again");
        return NBE_E_INVALID;
    }
    if (e->n == 0) return NBE_OK;
    const int rc = m->dist ? dist_step(e, nsteps) : multi_step_replicated(e, nsteps);
    if (rc != NBE_OK) m->poisoned = true;
    return rc;
}

static int multi_step_replicated(nbe_engine* e, int nsteps)
{
    MultiState* m = e->multi;
    const size_t G = m->subs.size();
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
