// C ABI of the gravity engine (see include/nbody_engine.h): engine lifetime, state upload/download and the
// step driver that strings the sort, build and force kernels together on one HIP stream.
#include "engine_internal.h"

#include <algorithm>
#include <cstdarg>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <new>
#include <vector>

static thread_local char g_err[512] = "";

void nbe_set_error(const char* fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* nbe_last_error(void) { return g_err; }

extern "C" int nbe_default_config(nbe_config* cfg)
{
    if (!cfg) { nbe_set_error("cfg is NULL"); return NBE_E_INVALID; }
    memset(cfg, 0, sizeof(*cfg));
    cfg->mode = NBE_MODE_DIRECT;
    cfg->precision = NBE_PREC_F32;
    cfg->G = 6.67408e-11;      // physics.go:132
    cfg->theta = 1.0;          // main.go:73
    cfg->dt = 3600.0;          // main.go:28
    cfg->eps = 0.0;
    cfg->bound_width = 65536.0;   // main.go:74-76
    cfg->device = 0;
    return NBE_OK;
}

static int use_device(nbe_engine* e)
{
    NBE_HIP(hipSetDevice(e->cfg.device));
    return NBE_OK;
}

// Calls that a multi-GPU engine does not offer; the rest are served by the coordinator or by the first GPU's engine.
#define NBE_NOT_MULTI(e, name)                                                                     \
    do {                                                                                           \
        if ((e)->multi) { nbe_set_error(name " is not available on an engine with num_devices > 1"); return NBE_E_UNSUPPORTED; } \
    } while (0)
#define NBE_FIRST(e) ((e) && (e)->multi ? multi_first(e) : (e))
// ... and those that an engine with cfg.distributed cannot serve from one GPU
#define NBE_NOT_DISTRIBUTED(e, name)                                                               \
    do {                                                                                           \
        if ((e) && multi_is_distributed(e)) { nbe_set_error(name " is not available on an engine with distributed = 1"); return NBE_E_UNSUPPORTED; } \
    } while (0)

static int free_state(nbe_engine* e)
{
    void** ptrs[] = {&e->posm[0], &e->posm[1], &e->vel[0], &e->vel[1], (void**)&e->mirror[0], (void**)&e->mirror[1],
                     (void**)&e->orig[0], (void**)&e->orig[1], &e->acc, &e->partial, (void**)&e->radius[0],
                     (void**)&e->radius[1]};
    for (void** p : ptrs)
        if (*p) { NBE_HIP(hipFree(*p)); *p = nullptr; }
    e->partial_len = 0;
    delete[] e->ids;
    e->ids = nullptr;
    e->n = 0;
    return NBE_OK;
}

extern "C" int nbe_create(const nbe_config* cfg, nbe_engine** out)
{
    if (!cfg || !out) { nbe_set_error("nbe_create: NULL argument"); return NBE_E_INVALID; }
    *out = nullptr;
    if (cfg->mode != NBE_MODE_DIRECT && cfg->mode != NBE_MODE_TREE) { nbe_set_error("bad mode %d", cfg->mode); return NBE_E_INVALID; }
    if (cfg->precision < NBE_PREC_F32 || cfg->precision > NBE_PREC_F64) { nbe_set_error("bad precision %d", cfg->precision); return NBE_E_INVALID; }
    if (!(cfg->bound_width > 0) || !(cfg->theta >= 0) || !(cfg->eps >= 0)) { nbe_set_error("bound_width must be > 0, theta and eps >= 0"); return NBE_E_INVALID; }
    if (cfg->collisions != 0 && cfg->collisions != 1) { nbe_set_error("collisions must be 0 or 1"); return NBE_E_INVALID; }
    if (cfg->reference_quirks & ~(NBE_QUIRK_REF_COM | NBE_QUIRK_DROP_TAIL)) { nbe_set_error("unknown bit in reference_quirks"); return NBE_E_INVALID; }
    if (cfg->walk_criterion != NBE_WALK_PER_BODY && cfg->walk_criterion != NBE_WALK_PER_WAVE) { nbe_set_error("bad walk_criterion %d", cfg->walk_criterion); return NBE_E_INVALID; }
    if (cfg->walk_criterion == NBE_WALK_PER_WAVE && (cfg->mode != NBE_MODE_TREE || cfg->reference_quirks)) {
        nbe_set_error("the per-wave walk criterion needs tree mode without reference_quirks");
        return NBE_E_INVALID;
    }
    if (cfg->reference_quirks && cfg->mode != NBE_MODE_TREE) { nbe_set_error("reference_quirks apply to tree mode only"); return NBE_E_INVALID; }
    if (cfg->num_devices < 0) { nbe_set_error("num_devices < 0"); return NBE_E_INVALID; }
    if (cfg->distributed != 0 && cfg->distributed != 1) { nbe_set_error("distributed must be 0 or 1"); return NBE_E_INVALID; }
    if (cfg->distributed && cfg->num_devices <= 1) { nbe_set_error("distributed = 1 needs num_devices > 1 (one GPU: use cell_sums = NBE_SUMS_EXACT for the same results)"); return NBE_E_INVALID; }
    if (cfg->cell_sums != NBE_SUMS_PREFIX && cfg->cell_sums != NBE_SUMS_TREE && cfg->cell_sums != NBE_SUMS_EXACT) { nbe_set_error("bad cell_sums %d", cfg->cell_sums); return NBE_E_INVALID; }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
        nbe_set_error("no HIP device available (this library has no CPU path)");
        return NBE_E_DEVICE;
    }
    if (cfg->num_devices > 1) {
        for (int r = 0; r < cfg->num_devices; r++) {
            const int d = cfg->device_ids ? cfg->device_ids[r] : r;
            if (d < 0 || d >= ndev) { nbe_set_error("device %d out of range (have %d)", d, ndev); return NBE_E_INVALID; }
        }
        return multi_create(cfg, out);
    }
    const int one_device = (cfg->num_devices == 1 && cfg->device_ids) ? cfg->device_ids[0] : cfg->device;
    if (one_device < 0 || one_device >= ndev) { nbe_set_error("device %d out of range (have %d)", one_device, ndev); return NBE_E_INVALID; }
    nbe_engine* e = new (std::nothrow) nbe_engine();
    if (!e) { nbe_set_error("out of host memory"); return NBE_E_NOMEM; }
    e->cfg = *cfg;
    e->cfg.device = one_device;
    e->cfg.num_devices = 1;
    e->cfg.device_ids = nullptr;               // the caller's array is not retained
    e->state64 = cfg->precision != NBE_PREC_F32;
    e->force64 = cfg->precision == NBE_PREC_F64;
    e->elem = e->state64 ? 32 : 16;
    e->small_max = tree_small_limit();
    if (const char* env = getenv("NBODY_SMALL_TREE_MAX")) {     // test knob: 0 sends every size down the tiled path
        const long v = strtol(env, nullptr, 10);
        if (v >= 0 && v < e->small_max) e->small_max = (int)v;
    }
    if (const char* env = getenv("NBODY_WALK_BODIES_PER_LANE")) {     // test knob: both fp32 walk kernels give the same bits
        const long v = strtol(env, nullptr, 10);
        if (v == 1 || v == 2) e->walk_bodies_per_lane = (int)v;
    }
    if (const char* env = getenv("NBODY_WALK_XCD_CHUNK_LOG2")) {      // measurement knob: how the walk's workgroups are dealt to the XCDs
        const long v = strtol(env, nullptr, 10);
        if (v >= 0 && v <= 12) e->walk_xcd_shift = (int)v;
    }
    if (const char* env = getenv("NBODY_WALK_PRIVATE")) e->walk_private = env[0] == '1';     // measurement knob, see walk_private_kernel
    if (const char* env = getenv("NBODY_WALK_SPARSE")) {             // measurement knob: bodies per wavefront of small launches
        const long v = strtol(env, nullptr, 10);
        if (v >= 0 && v <= 6) e->walk_sparse = (int)v;
    }
    int rc = use_device(e);
    if (rc == NBE_OK && hipStreamCreateWithFlags(&e->own_stream, hipStreamNonBlocking) != hipSuccess) {
        nbe_set_error("hipStreamCreate failed");
        rc = NBE_E_DEVICE;
    }
    for (int k = 0; k < 5 && rc == NBE_OK; k++)
        if (hipEventCreate(&e->ev[k]) != hipSuccess) { nbe_set_error("hipEventCreate failed"); rc = NBE_E_DEVICE; }
    if (rc != NBE_OK) {
        for (int k = 0; k < 5; k++)
            if (e->ev[k]) (void)hipEventDestroy(e->ev[k]);
        if (e->own_stream) (void)hipStreamDestroy(e->own_stream);
        delete e;
        return rc;
    }
    e->stream = e->own_stream;
    *out = e;
    return NBE_OK;
}

extern "C" void nbe_destroy(nbe_engine* e)
{
    if (!e) return;
    if (e->multi) { multi_destroy(e); return; }
    (void)hipSetDevice(e->cfg.device);
    dist_destroy(e);
    (void)frame_stream_stop(e);
    (void)hipStreamSynchronize(e->stream);
    (void)free_state(e);
    (void)collide_workspace_free(e);
    (void)direct_collide_free(e);
    TreeWork& w = e->tw;
    void* ptrs[] = {w.keys[0], w.keys[1], w.sidx[0], w.sidx[1], w.block_hist, w.scan_tmp, w.bt_first, w.bt_last, w.bt_parent,
                    w.bt_level, w.bt_cell, w.lvlmask, w.cellcnt, w.nodes, w.node_level, w.node_sum, w.counters, w.d_interactions,
                    w.prefix, w.prefix_tiles, w.prefixq, w.prefixq_tiles, e->d_energy};
    for (void* p : ptrs)
        if (p) (void)hipFree(p);
    for (int k = 0; k < 5; k++)
        if (e->ev[k]) (void)hipEventDestroy(e->ev[k]);
    if (e->own_stream) (void)hipStreamDestroy(e->own_stream);
    delete e;
}

extern "C" int nbe_set_stream(nbe_engine* e, void* hip_stream)
{
    if (!e) { nbe_set_error("engine is NULL"); return NBE_E_INVALID; }
    NBE_NOT_MULTI(e, "nbe_set_stream");
    NBE_TRY(use_device(e));
    NBE_HIP(hipStreamSynchronize(e->stream));
    e->stream = (hipStream_t)hip_stream;
    return NBE_OK;
}

extern "C" int nbe_upload(nbe_engine* e, size_t n, const double* pos, const double* vel, const double* mass, const uint64_t* id)
{
    if (!e) { nbe_set_error("engine is NULL"); return NBE_E_INVALID; }
    if (n > 0 && (!pos || !vel || !mass)) { nbe_set_error("nbe_upload: NULL array"); return NBE_E_INVALID; }
    if (n > 0x7fffffffull) { nbe_set_error("at most 2^31-1 bodies"); return NBE_E_UNSUPPORTED; }
    // (with distributed = 1 no GPU holds all bodies: the limit is 2^26 per GPU there, checked by nbe_dist_upload)
    if (e->cfg.mode == NBE_MODE_TREE && n > (1ull << 26) && !multi_is_distributed(e)) { nbe_set_error("tree mode supports at most 2^26 bodies"); return NBE_E_UNSUPPORTED; }
    for (size_t i = 0; i < n; i++)
        if (!(mass[i] >= 0.0)) { nbe_set_error("mass[%zu] is negative or NaN", i); return NBE_E_INVALID; }
    if (e->multi) return multi_upload(e, n, pos, vel, mass, id);
    int rc;
    if (e->dist) { nbe_set_error("nbe_upload on a distributed engine; use nbe_dist_upload"); return NBE_E_INVALID; }
    try { rc = engine_upload_state(e, n, pos, vel, mass, id); }
    catch (const std::bad_alloc&) { nbe_set_error("out of host memory"); rc = NBE_E_NOMEM; }
    if (rc != NBE_OK) {
        // leave an empty engine behind rather than half of a state (the message of the failure is kept)
        char msg[512];
        snprintf(msg, sizeof(msg), "%s", g_err);
        (void)free_state(e);
        e->stats.n = 0; e->stats.n_alive = 0; e->host_nalive = 0; e->shard_count = 0;
        nbe_set_error("%s", msg);
    }
    return rc;
}

int engine_upload_state(nbe_engine* e, size_t n, const double* pos, const double* vel, const double* mass, const uint64_t* id)
{
    NBE_TRY(use_device(e));
    NBE_TRY(frame_stream_stop(e));
    NBE_HIP(hipStreamSynchronize(e->stream));
    NBE_TRY(free_state(e));
    e->n = n;
    e->cur = 0; e->orig_cur = 0;
    e->prepared = false; e->advance_pending = false; e->advance_parts = 0; e->have_counts = false; e->keys_sorted = false;
    e->shard_set = false; e->shard_first = 0; e->shard_count = n;
    memset(&e->stats, 0, sizeof(e->stats));
    e->stats.n = n; e->stats.n_alive = n;
    e->host_nalive = n;
    // Equal masses let the fp32 direct sum leave the mass out of its pair loop (direct.hip).  The value the kernels
    // hold is the rounded one in fp32 state; masses change only through collisions, which keep the general kernel.
    e->equal_mass = 0.0;
    if (n > 0 && e->cfg.mode == NBE_MODE_DIRECT && !e->cfg.collisions && mass[0] > 0.0 && getenv("NBODY_NO_EQUAL_MASS") == nullptr) {
        bool same = true;
        for (size_t i = 1; i < n && same; i++) same = mass[i] == mass[0];
        const double held = e->cfg.precision == NBE_PREC_F64 ? mass[0] : (double)(float)mass[0];
        if (same && held > 0.0 && std::isfinite(held)) e->equal_mass = held;
    }
    // the quanta of exact cell sums (NBE_SUMS_EXACT) follow from the total mass, which collisions keep and bodies that
    // leave the bound only lower (fp32 state holds rounded masses: a little headroom covers their sum)
    {
        double total = 0.0;
        for (size_t i = 0; i < n; i++) total += mass[i];
        tree_set_total_mass(e, total * (1.0 + 0x1p-20));
    }
    const size_t cap = n > 0 ? n : 1;
    for (int b = 0; b < 2; b++) {
        NBE_HIP(hipMalloc(&e->posm[b], cap * e->elem));
        NBE_HIP(hipMalloc(&e->vel[b], cap * e->elem));
        NBE_HIP(hipMalloc((void**)&e->orig[b], cap * 4));
        if (e->cfg.precision == NBE_PREC_MIXED) NBE_HIP(hipMalloc((void**)&e->mirror[b], cap * sizeof(float4)));
    }
    NBE_HIP(hipMalloc(&e->acc, cap * e->elem));
    if (e->cfg.collisions) {
        NBE_HIP(hipMalloc((void**)&e->radius[0], cap * sizeof(double)));
        NBE_HIP(hipMalloc((void**)&e->radius[1], cap * sizeof(double)));
    }
    e->ids = new (std::nothrow) uint64_t[cap];
    if (!e->ids) { nbe_set_error("out of host memory"); return NBE_E_NOMEM; }
    for (size_t i = 0; i < n; i++) e->ids[i] = id ? id[i] : (uint64_t)i;
    if (n == 0) return NBE_OK;

    std::vector<uint32_t> ident(n);
    for (size_t i = 0; i < n; i++) ident[i] = (uint32_t)i;
    if (e->state64) {
        std::vector<double> hp(4 * n), hv(4 * n);
        for (size_t i = 0; i < n; i++) {
            hp[4 * i] = pos[3 * i]; hp[4 * i + 1] = pos[3 * i + 1]; hp[4 * i + 2] = pos[3 * i + 2]; hp[4 * i + 3] = mass[i];
            hv[4 * i] = vel[3 * i]; hv[4 * i + 1] = vel[3 * i + 1]; hv[4 * i + 2] = vel[3 * i + 2]; hv[4 * i + 3] = 0.0;
        }
        NBE_HIP(hipMemcpy(e->posm[0], hp.data(), n * 32, hipMemcpyHostToDevice));
        NBE_HIP(hipMemcpy(e->vel[0], hv.data(), n * 32, hipMemcpyHostToDevice));
        if (e->cfg.precision == NBE_PREC_MIXED) {
            std::vector<float> hm(4 * n);
            for (size_t i = 0; i < 4 * n; i++) hm[i] = (float)hp[i];
            NBE_HIP(hipMemcpy(e->mirror[0], hm.data(), n * 16, hipMemcpyHostToDevice));
        }
    } else {
        std::vector<float> hp(4 * n), hv(4 * n);
        for (size_t i = 0; i < n; i++) {
            hp[4 * i] = (float)pos[3 * i]; hp[4 * i + 1] = (float)pos[3 * i + 1]; hp[4 * i + 2] = (float)pos[3 * i + 2];
            hp[4 * i + 3] = (float)mass[i];
            hv[4 * i] = (float)vel[3 * i]; hv[4 * i + 1] = (float)vel[3 * i + 1]; hv[4 * i + 2] = (float)vel[3 * i + 2];
            hv[4 * i + 3] = 0.0f;
        }
        NBE_HIP(hipMemcpy(e->posm[0], hp.data(), n * 16, hipMemcpyHostToDevice));
        NBE_HIP(hipMemcpy(e->vel[0], hv.data(), n * 16, hipMemcpyHostToDevice));
    }
    NBE_HIP(hipMemcpy(e->orig[0], ident.data(), n * 4, hipMemcpyHostToDevice));
    NBE_HIP(hipMemcpy(e->orig[1], ident.data(), n * 4, hipMemcpyHostToDevice));
    if (e->cfg.mode == NBE_MODE_TREE) NBE_TRY(tree_workspace_reserve(e));
    if (e->cfg.collisions) {
        NBE_TRY(collide_workspace_reserve(e));
        e->cw.begun = false;
        NBE_TRY(collide_fill_radius(e, e->radius[0], 2.0));      // physics.go:20 defaultRadius
        NBE_HIP(hipMemsetAsync(e->cw.dead, 0, n, e->stream));
        if (e->cfg.mode == NBE_MODE_TREE) {
            NBE_HIP(hipMemsetAsync(e->tw.counters, 0, 16 * sizeof(int), e->stream));
        } else {
            NBE_TRY(direct_collide_reserve(e));
            NBE_HIP(hipMemsetAsync(e->dc_counters, 0, 8 * sizeof(int), e->stream));
            e->direct_absorbed = 0;
        }
    }
    return NBE_OK;
}

// Device slots [0, *slots) held the live bodies at the last build (n until a build has removed some; device
// order keeps the removed ones at the tail).  Collisions absorb bodies between builds: those are flagged in
// e->cw.dead and counted out of stats.n_alive.
static int read_counts(nbe_engine* e, size_t* slots)
{
    size_t* n_alive = slots;
    size_t absorbed = 0;
    *n_alive = e->n;
    // (a rank of the distributed tree that holds no body at the moment still has its count of absorbed bodies)
    if (e->have_counts && (e->n > 0 || e->dist)) {
        int c[8];
        NBE_HIP(hipMemcpyAsync(c, e->tw.counters, sizeof(c), hipMemcpyDeviceToHost, e->stream));
        NBE_HIP(hipStreamSynchronize(e->stream));
        *n_alive = (size_t)c[0];
        absorbed = (size_t)c[3];
        e->stats.tree_nodes = (uint64_t)c[1];
        e->stats.tree_cells = (uint64_t)c[2];
        e->stats.merges = (uint64_t)c[6];
    }
    if (e->cfg.collisions && e->cfg.mode == NBE_MODE_DIRECT) {
        absorbed = e->direct_absorbed;
        e->stats.merges = e->direct_absorbed;
    }
    e->host_nalive = *n_alive - absorbed;
    e->stats.n_alive = *n_alive - absorbed;
    return NBE_OK;
}

// Host copy of the absorbed flags of the first `slots` device slots (all zero when collisions are off).
static int read_dead(nbe_engine* e, size_t slots, std::vector<uint8_t>* dead)
{
    dead->assign(slots, 0);
    if (e->cfg.collisions && (e->have_counts || e->cfg.mode == NBE_MODE_DIRECT) && slots > 0) {
        NBE_HIP(hipMemcpyAsync(dead->data(), e->cw.dead, slots, hipMemcpyDeviceToHost, e->stream));
        NBE_HIP(hipStreamSynchronize(e->stream));
    }
    return NBE_OK;
}

// Radii follow the state buffers in tree mode (they are reordered with it); the direct sum never reorders.
static double* radius_buf(nbe_engine* e) { return e->cfg.mode == NBE_MODE_TREE ? e->radius[e->cur] : e->radius[0]; }

// upload index -> device slot of every live body (0xffffffff for removed ones).
static int live_slots(nbe_engine* e, std::vector<uint32_t>* slot_of, size_t* n_alive)
{
    const size_t n = e->n;
    size_t slots = 0;
    NBE_TRY(read_counts(e, &slots));
    std::vector<uint8_t> dead;
    NBE_TRY(read_dead(e, slots, &dead));
    std::vector<uint32_t> horig(n);
    if (n > 0) {
        NBE_HIP(hipMemcpyAsync(horig.data(), e->orig[e->orig_cur], n * 4, hipMemcpyDeviceToHost, e->stream));
        NBE_HIP(hipStreamSynchronize(e->stream));
    }
    slot_of->assign(n, 0xffffffffu);
    size_t k = 0;
    for (size_t d = 0; d < slots; d++)
        if (!dead[d]) { (*slot_of)[horig[d]] = (uint32_t)d; k++; }
    *n_alive = k;
    return NBE_OK;
}

static int download_impl(nbe_engine* e, double* pos, double* vel, double* mass, uint64_t* id, size_t* n_alive_out)
{
    if (!e) { nbe_set_error("engine is NULL"); return NBE_E_INVALID; }
    NBE_TRY(use_device(e));
    if (e->advance_pending) { nbe_set_error("nbe_download between nbe_shard_advance and nbe_shard_commit"); return NBE_E_INVALID; }
    const size_t n = e->n;
    std::vector<uint32_t> slot_of;
    size_t na = 0;
    NBE_TRY(live_slots(e, &slot_of, &na));
    if (n_alive_out) *n_alive_out = na;
    if (n == 0) return NBE_OK;
    std::vector<double> hp, hv;
    std::vector<float> fp, fv;
    if (e->state64) {
        hp.resize(4 * n); hv.resize(4 * n);
        NBE_HIP(hipMemcpyAsync(hp.data(), e->posm[e->cur], n * 32, hipMemcpyDeviceToHost, e->stream));
        NBE_HIP(hipMemcpyAsync(hv.data(), e->vel[e->cur], n * 32, hipMemcpyDeviceToHost, e->stream));
    } else {
        fp.resize(4 * n); fv.resize(4 * n);
        NBE_HIP(hipMemcpyAsync(fp.data(), e->posm[e->cur], n * 16, hipMemcpyDeviceToHost, e->stream));
        NBE_HIP(hipMemcpyAsync(fv.data(), e->vel[e->cur], n * 16, hipMemcpyDeviceToHost, e->stream));
    }
    NBE_HIP(hipStreamSynchronize(e->stream));
    size_t k = 0;                                  // survivors in upload order
    for (size_t u = 0; u < n; u++) {
        const uint32_t d = slot_of[u];
        if (d == 0xffffffffu) continue;
        for (int c = 0; c < 3; c++) {
            if (pos) pos[3 * k + c] = e->state64 ? hp[4 * (size_t)d + c] : (double)fp[4 * (size_t)d + c];
            if (vel) vel[3 * k + c] = e->state64 ? hv[4 * (size_t)d + c] : (double)fv[4 * (size_t)d + c];
        }
        if (mass) mass[k] = e->state64 ? hp[4 * (size_t)d + 3] : (double)fp[4 * (size_t)d + 3];
        if (id) id[k] = e->ids[u];
        k++;
    }
    return NBE_OK;
}

static int set_radius_impl(nbe_engine* e, const double* radius)
{
    if (!e) { nbe_set_error("engine is NULL"); return NBE_E_INVALID; }
    if (!e->cfg.collisions) { nbe_set_error("nbe_set_radius needs an engine created with collisions on"); return NBE_E_INVALID; }
    if (e->n > 0 && !radius) { nbe_set_error("radius is NULL"); return NBE_E_INVALID; }
    NBE_TRY(use_device(e));
    const size_t n = e->n;
    if (n == 0) return NBE_OK;
    for (size_t u = 0; u < n; u++)
        if (!(radius[u] >= 0.0)) { nbe_set_error("radius[%zu] is negative or NaN", u); return NBE_E_INVALID; }
    std::vector<uint32_t> horig(n);
    NBE_HIP(hipMemcpyAsync(horig.data(), e->orig[e->orig_cur], n * 4, hipMemcpyDeviceToHost, e->stream));
    NBE_HIP(hipStreamSynchronize(e->stream));
    std::vector<double> dev(n);
    for (size_t d = 0; d < n; d++) dev[d] = radius[horig[d]];
    NBE_HIP(hipMemcpy(radius_buf(e), dev.data(), n * sizeof(double), hipMemcpyHostToDevice));
    return NBE_OK;
}

static int get_radius_impl(nbe_engine* e, double* radius, size_t* n_alive_out)
{
    if (!e) { nbe_set_error("engine is NULL"); return NBE_E_INVALID; }
    if (!e->cfg.collisions) { nbe_set_error("nbe_get_radius needs an engine created with collisions on"); return NBE_E_INVALID; }
    NBE_TRY(use_device(e));
    const size_t n = e->n;
    std::vector<uint32_t> slot_of;
    size_t na = 0;
    NBE_TRY(live_slots(e, &slot_of, &na));
    if (n_alive_out) *n_alive_out = na;
    if (n == 0 || !radius) return NBE_OK;
    std::vector<double> dev(n);
    NBE_HIP(hipMemcpyAsync(dev.data(), radius_buf(e), n * sizeof(double), hipMemcpyDeviceToHost, e->stream));
    NBE_HIP(hipStreamSynchronize(e->stream));
    size_t k = 0;
    for (size_t u = 0; u < n; u++)
        if (slot_of[u] != 0xffffffffu) radius[k++] = dev[slot_of[u]];
    return NBE_OK;
}

static int force_pass(nbe_engine* e, uint64_t t0, uint64_t t1, bool fused, bool count = false)
{
    if (fused) NBE_TRY(frame_stream_guard_write(e, e->cur ^ 1));
    if (e->cfg.mode == NBE_MODE_TREE) return tree_forces(e, e->cur, t0, t1, fused, count, e->cur ^ 1);
    return direct_forces(e, e->cur, t0, t1, fused, e->cur ^ 1);
}

// Sharded stepping with collisions on (tree mode only): the force pass, which is nearly all of a step, is what the
// engines share out; each computes the accelerations of its range into the acceleration buffer, the host exchanges
// those, and nbe_shard_commit then runs the collision search, the merges and the update (collide.hip) over all bodies
// on every engine alike, so that the replicated states stay identical.  The direct sum with collisions detects
// touching pairs over all pairs and replays a merging step serially, which does not shard.
static int check_sharded_collisions(const nbe_engine* e)
{
    if (e->dist) { nbe_set_error("nbe_shard_* on a distributed engine; use the nbe_dist_* calls"); return NBE_E_INVALID; }
    if (e->cfg.collisions && e->cfg.mode != NBE_MODE_TREE) {
        nbe_set_error("sharded stepping supports collisions in tree mode only; use nbe_step");
        return NBE_E_UNSUPPORTED;
    }
    return NBE_OK;
}

extern "C" int nbe_shard_prepare(nbe_engine* e)
{
    if (!e) { nbe_set_error("engine is NULL"); return NBE_E_INVALID; }
    NBE_NOT_MULTI(e, "nbe_shard_prepare");
    NBE_TRY(check_sharded_collisions(e));
    NBE_TRY(use_device(e));
    if (e->advance_pending) { nbe_set_error("nbe_shard_prepare before nbe_shard_commit"); return NBE_E_INVALID; }
    NBE_TRY(frame_stream_before_step(e));       // sends this step's frame once, however often prepare is called
    if (e->cfg.mode == NBE_MODE_TREE && !e->prepared) NBE_TRY(tree_sort_and_build(e));
    e->prepared = true;
    return NBE_OK;
}

extern "C" int nbe_shard_sort(nbe_engine* e)
{
    if (!e) { nbe_set_error("engine is NULL"); return NBE_E_INVALID; }
    NBE_NOT_MULTI(e, "nbe_shard_sort");
    NBE_TRY(check_sharded_collisions(e));
    NBE_TRY(use_device(e));
    if (e->advance_pending) { nbe_set_error("nbe_shard_sort before nbe_shard_commit"); return NBE_E_INVALID; }
    if (e->cfg.mode != NBE_MODE_TREE || e->prepared || e->cfg.collisions) return NBE_OK;   // (collisions: nothing runs ahead)
    return tree_sort_keys(e);
}

extern "C" int nbe_shard_advance(nbe_engine* e)
{
    if (!e) { nbe_set_error("engine is NULL"); return NBE_E_INVALID; }
    NBE_NOT_MULTI(e, "nbe_shard_advance");
    NBE_TRY(check_sharded_collisions(e));
    NBE_TRY(use_device(e));
    if (!e->prepared) { nbe_set_error("nbe_shard_advance without nbe_shard_prepare"); return NBE_E_INVALID; }
    if (e->advance_pending) { nbe_set_error("nbe_shard_advance called twice (advance several ranges with nbe_shard_advance_range)"); return NBE_E_INVALID; }
    uint64_t t0 = e->shard_set ? e->shard_first : 0;
    uint64_t t1 = e->shard_set ? e->shard_first + e->shard_count : e->n;
    if (t1 > e->n) t1 = e->n;
    if (t0 > t1) t0 = t1;
    NBE_TRY(force_pass(e, t0, t1, !e->cfg.collisions));      // collisions: accelerations only, the update follows the merges
    e->advance_pending = true;
    e->advance_parts++;
    return NBE_OK;
}

extern "C" int nbe_shard_advance_range(nbe_engine* e, uint64_t first, uint64_t count)
{
    if (!e) { nbe_set_error("engine is NULL"); return NBE_E_INVALID; }
    NBE_NOT_MULTI(e, "nbe_shard_advance_range");
    NBE_TRY(check_sharded_collisions(e));
    NBE_TRY(use_device(e));
    if (!e->prepared) { nbe_set_error("nbe_shard_advance_range without nbe_shard_prepare"); return NBE_E_INVALID; }
    if (first > e->n || count > e->n - first) { nbe_set_error("range [%llu, +%llu) outside 0..%zu", (unsigned long long)first, (unsigned long long)count, e->n); return NBE_E_INVALID; }
    if (count > 0) NBE_TRY(force_pass(e, first, first + count, !e->cfg.collisions));
    e->advance_pending = true;
    e->advance_parts += 2;                       // the ranges advanced are not (only) the shard set with nbe_set_shard
    return NBE_OK;
}

// MIXED precision keeps an fp32 copy of the positions for the force kernels.  An advance refreshes it only
// for the engine's own shard; rows that arrive from other engines through the caller's all-gather are
// converted here.
__global__ __launch_bounds__(256) void refresh_mirror_kernel(const double4* __restrict__ posm, float4* __restrict__ mirror,
                                                             int n, int skip_first, int skip_end)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n || (i >= skip_first && i < skip_end)) return;
    const double4 p = posm[i];
    mirror[i] = make_float4((float)p.x, (float)p.y, (float)p.z, (float)p.w);
}

extern "C" int nbe_shard_commit(nbe_engine* e)
{
    if (!e) { nbe_set_error("engine is NULL"); return NBE_E_INVALID; }
    NBE_NOT_MULTI(e, "nbe_shard_commit");
    if (!e->advance_pending) { nbe_set_error("nbe_shard_commit without nbe_shard_advance"); return NBE_E_INVALID; }
    NBE_TRY(use_device(e));
    const bool by_ranges = e->advance_parts > 1;      // nbe_shard_advance_range: rows of any range may have arrived from elsewhere
    if (e->cfg.collisions) {
        // every engine holds all accelerations now: search, merge and update all bodies (writes the next-state buffers)
        NBE_TRY(collide_and_update(e));
    } else if (e->cfg.precision == NBE_PREC_MIXED && (by_ranges || (e->shard_set && e->shard_count < e->n))) {
        const int nxt = e->cur ^ 1;
        // with explicit ranges every row is converted (the engine's own rows come out as the update wrote them)
        hipLaunchKernelGGL(refresh_mirror_kernel, dim3(cdiv(e->n, 256)), dim3(256), 0, e->stream,
                           (const double4*)e->posm[nxt], e->mirror[nxt], (int)e->n, by_ranges ? 0 : (int)e->shard_first,
                           by_ranges ? 0 : (int)(e->shard_first + e->shard_count));
        NBE_HIP(hipGetLastError());
    }
    e->cur ^= 1;
    e->advance_parts = 0;
    e->advance_pending = false;
    e->prepared = false;
    e->stats.steps++;
    return NBE_OK;
}

extern "C" int nbe_set_shard(nbe_engine* e, uint64_t first, uint64_t count)
{
    if (!e) { nbe_set_error("engine is NULL"); return NBE_E_INVALID; }
    NBE_NOT_MULTI(e, "nbe_set_shard");
    if (first > e->n || count > e->n - first) { nbe_set_error("shard [%llu, +%llu) outside 0..%zu", (unsigned long long)first, (unsigned long long)count, e->n); return NBE_E_INVALID; }
    if (e->advance_pending) { nbe_set_error("nbe_set_shard between advance and commit"); return NBE_E_INVALID; }
    e->shard_first = first;
    e->shard_count = count;
    e->shard_set = true;
    return NBE_OK;
}

extern "C" int nbe_device_state(nbe_engine* e, nbe_device_state_t* out)
{
    if (!e || !out) { nbe_set_error("NULL argument"); return NBE_E_INVALID; }
    NBE_NOT_MULTI(e, "nbe_device_state");
    out->pos_mass = e->posm[e->cur];
    out->vel = e->vel[e->cur];
    out->pos_mass_next = e->posm[e->cur ^ 1];
    out->vel_next = e->vel[e->cur ^ 1];
    out->n = e->n;
    out->elem_bytes = (uint32_t)e->elem;
    out->reserved = 0;
    out->acc = e->acc;
    return NBE_OK;
}

static int step_once_full(nbe_engine* e)
{
    NBE_TRY(frame_stream_before_step(e));
    if (e->cfg.mode == NBE_MODE_TREE && !e->prepared) NBE_TRY(tree_sort_and_build(e));
    else if (e->timing) {
        NBE_HIP(hipEventRecord(e->ev[0], e->stream));
        NBE_HIP(hipEventRecord(e->ev[1], e->stream));
        NBE_HIP(hipEventRecord(e->ev[2], e->stream));
    }
    if (e->timing) {
        // the production walk does not count; a separate counting pass (outside the timed span) fills the stats
        if (e->cfg.mode == NBE_MODE_TREE) NBE_TRY(force_pass(e, 0, e->n, false, true));
        NBE_HIP(hipEventRecord(e->ev[4], e->stream));
    }
    if (e->cfg.collisions && e->cfg.mode == NBE_MODE_DIRECT) {
        // a step with a touching pair is replayed in the reference's order of events; any other is the fused step
        bool replayed = false;
        NBE_TRY(direct_collide_step(e, &replayed));
        if (!replayed) NBE_TRY(force_pass(e, 0, e->n, true));
    } else if (e->cfg.collisions) {
        // accelerations first (the collision merge needs them apart from the update), then search, merge, update
        NBE_TRY(force_pass(e, 0, e->n, false));
        NBE_TRY(collide_and_update(e));
    } else {
        NBE_TRY(force_pass(e, 0, e->n, true));
    }
    if (e->timing) NBE_HIP(hipEventRecord(e->ev[3], e->stream));
    e->cur ^= 1;
    e->prepared = false;
    e->stats.steps++;
    return NBE_OK;
}

extern "C" int nbe_step(nbe_engine* e, int nsteps)
{
    if (!e) { nbe_set_error("engine is NULL"); return NBE_E_INVALID; }
    if (nsteps < 0) { nbe_set_error("nsteps < 0"); return NBE_E_INVALID; }
    if (e->multi) return multi_step(e, nsteps);
    if (e->dist) { nbe_set_error("nbe_step on a distributed engine; use the nbe_dist_* calls"); return NBE_E_INVALID; }
    NBE_TRY(use_device(e));
    if (e->advance_pending) { nbe_set_error("nbe_step between nbe_shard_advance and nbe_shard_commit"); return NBE_E_INVALID; }
    if (e->shard_set && (e->shard_first != 0 || e->shard_count != e->n)) {
        nbe_set_error("nbe_step on an engine with a partial shard; use the nbe_shard_* calls");
        return NBE_E_INVALID;
    }
    if (e->n == 0) return NBE_OK;
    for (int s = 0; s < nsteps; s++) NBE_TRY(step_once_full(e));
    return NBE_OK;
}

static int fetch_interactions(nbe_engine* e)
{
    if (e->cfg.mode == NBE_MODE_TREE && e->tw.d_interactions) {
        unsigned long long v[8] = {0};
        NBE_HIP(hipMemcpyAsync(v, e->tw.d_interactions, sizeof(v), hipMemcpyDeviceToHost, e->stream));
        NBE_HIP(hipStreamSynchronize(e->stream));
        e->stats.interactions = v[0];
        if (e->walk_private && e->cfg.precision == NBE_PREC_F32) fprintf(stderr, "one-body-one-cursor walk: %llu node visits, %llu force evaluations\n", v[1], v[0]);
#ifdef WALK_VISIT_STATS
        fprintf(stderr, "walk visits: sum %llu max %llu sumsq %llu hist(<3000,<4000,<6000,>=6000) %llu %llu %llu %llu\n", v[1], v[2],
                v[3], v[4], v[5], v[6], v[7]);
#endif
    }
    return NBE_OK;
}

extern "C" int nbe_step_timed(nbe_engine* e, nbe_stats_t* stats)
{
    if (!e) { nbe_set_error("engine is NULL"); return NBE_E_INVALID; }
    NBE_NOT_MULTI(e, "nbe_step_timed");
    NBE_TRY(use_device(e));
    if (e->n == 0) { if (stats) *stats = e->stats; return NBE_OK; }
    if (e->advance_pending) { nbe_set_error("nbe_step_timed between nbe_shard_advance and nbe_shard_commit"); return NBE_E_INVALID; }
    // a tree left by nbe_accel, nbe_energy or nbe_tree_nodes is reused: sort and build then read as 0 ms
    e->timing = true;
    int rc = step_once_full(e);
    e->timing = false;
    NBE_TRY(rc);
    NBE_HIP(hipEventSynchronize(e->ev[3]));
    NBE_HIP(hipEventElapsedTime(&e->stats.ms_sort, e->ev[0], e->ev[1]));
    NBE_HIP(hipEventElapsedTime(&e->stats.ms_build, e->ev[1], e->ev[2]));
    NBE_HIP(hipEventElapsedTime(&e->stats.ms_force, e->ev[4], e->ev[3]));
    e->stats.ms_total = e->stats.ms_sort + e->stats.ms_build + e->stats.ms_force;
    NBE_TRY(fetch_interactions(e));
    size_t na;
    NBE_TRY(read_counts(e, &na));
    if (stats) *stats = e->stats;
    return NBE_OK;
}

extern "C" int nbe_sync(nbe_engine* e)
{
    if (!e) { nbe_set_error("engine is NULL"); return NBE_E_INVALID; }
    if (e->multi) return multi_sync(e);
    NBE_TRY(use_device(e));
    NBE_HIP(hipStreamSynchronize(e->stream));
    return NBE_OK;
}

extern "C" int nbe_stats(nbe_engine* e, nbe_stats_t* stats)
{
    if (!e || !stats) { nbe_set_error("NULL argument"); return NBE_E_INVALID; }
    if (multi_is_distributed(e)) return multi_dist_stats(e, stats);
    e = NBE_FIRST(e);
    NBE_TRY(use_device(e));
    NBE_TRY(fetch_interactions(e));
    size_t na;
    NBE_TRY(read_counts(e, &na));
    e->stats.n = e->n;
    *stats = e->stats;
    return NBE_OK;
}

static int accel_impl(nbe_engine* e, double* acc_xyz, uint8_t* alive_out)
{
    if (!e) { nbe_set_error("engine is NULL"); return NBE_E_INVALID; }
    NBE_TRY(use_device(e));
    if (e->advance_pending) { nbe_set_error("nbe_accel between nbe_shard_advance and nbe_shard_commit"); return NBE_E_INVALID; }
    const size_t n = e->n;
    if (n == 0) return NBE_OK;
    if (!acc_xyz) { nbe_set_error("acc_xyz is NULL"); return NBE_E_INVALID; }
    NBE_HIP(hipMemsetAsync(e->acc, 0, n * e->elem, e->stream));
    if (e->cfg.mode == NBE_MODE_TREE && !e->prepared) { NBE_TRY(tree_sort_and_build(e)); e->prepared = true; }
    NBE_TRY(force_pass(e, 0, n, false, true));
    std::vector<uint32_t> horig(n);
    NBE_HIP(hipMemcpyAsync(horig.data(), e->orig[e->orig_cur], n * 4, hipMemcpyDeviceToHost, e->stream));
    std::vector<double> ha;
    std::vector<float> fa;
    if (e->state64) { ha.resize(4 * n); NBE_HIP(hipMemcpyAsync(ha.data(), e->acc, n * 32, hipMemcpyDeviceToHost, e->stream)); }
    else { fa.resize(4 * n); NBE_HIP(hipMemcpyAsync(fa.data(), e->acc, n * 16, hipMemcpyDeviceToHost, e->stream)); }
    NBE_HIP(hipStreamSynchronize(e->stream));
    NBE_TRY(fetch_interactions(e));
    size_t na;
    NBE_TRY(read_counts(e, &na));
    std::vector<uint8_t> dead;
    NBE_TRY(read_dead(e, na, &dead));
    for (size_t d = 0; d < n; d++) {
        const size_t u = horig[d];
        const bool live = d < na && !dead[d];
        for (int c = 0; c < 3; c++)
            acc_xyz[3 * u + c] = live ? (e->state64 ? ha[4 * d + c] : (double)fa[4 * d + c]) : 0.0;
        if (alive_out) alive_out[u] = live ? 1 : 0;
    }
    return NBE_OK;
}

extern "C" int nbe_energy(nbe_engine* e, double* kinetic, double* potential, double momentum[3])
{
    if (!e) { nbe_set_error("engine is NULL"); return NBE_E_INVALID; }
    if (multi_is_distributed(e)) return multi_dist_energy(e, kinetic, potential, momentum);
    e = NBE_FIRST(e);
    NBE_TRY(use_device(e));
    if (e->advance_pending) { nbe_set_error("nbe_energy between nbe_shard_advance and nbe_shard_commit"); return NBE_E_INVALID; }
    return energy_compute(e, kinetic, potential, momentum);
}

extern "C" int nbe_time_force_kernel(nbe_engine* e, int reps, float* ms_per_launch)
{
    if (!e || !ms_per_launch || reps <= 0) { nbe_set_error("bad argument"); return NBE_E_INVALID; }
    NBE_NOT_DISTRIBUTED(e, "nbe_time_force_kernel");
    e = NBE_FIRST(e);          // several GPUs: the first one's range
    NBE_TRY(use_device(e));
    if (e->advance_pending) { nbe_set_error("nbe_time_force_kernel inside a sharded step"); return NBE_E_INVALID; }
    *ms_per_launch = 0;
    if (e->n == 0) return NBE_OK;
    if (e->cfg.mode == NBE_MODE_TREE && !e->prepared) { NBE_TRY(tree_sort_and_build(e)); e->prepared = true; }
    uint64_t t0 = e->shard_set ? e->shard_first : 0;
    uint64_t t1 = e->shard_set ? e->shard_first + e->shard_count : e->n;
    NBE_TRY(force_pass(e, t0, t1, false, true));   // warm-up; also counts the interactions of one pass
    NBE_HIP(hipEventRecord(e->ev[0], e->stream));
    for (int r = 0; r < reps; r++) NBE_TRY(force_pass(e, t0, t1, false));
    NBE_HIP(hipEventRecord(e->ev[3], e->stream));
    NBE_HIP(hipEventSynchronize(e->ev[3]));
    float ms = 0;
    NBE_HIP(hipEventElapsedTime(&ms, e->ev[0], e->ev[3]));
    *ms_per_launch = ms / (float)reps;
    NBE_TRY(fetch_interactions(e));
    return NBE_OK;
}

static int tree_nodes_impl(nbe_engine* e, double* out, size_t cap, size_t* count)
{
    if (!e || !count) { nbe_set_error("NULL argument"); return NBE_E_INVALID; }
    NBE_TRY(use_device(e));
    *count = 0;
    if (e->cfg.mode != NBE_MODE_TREE) { nbe_set_error("nbe_tree_nodes needs tree mode"); return NBE_E_INVALID; }
    if (e->advance_pending) { nbe_set_error("nbe_tree_nodes inside a sharded step"); return NBE_E_INVALID; }
    if (e->n == 0) return NBE_OK;
    if (!e->prepared) { NBE_TRY(tree_sort_and_build(e)); e->prepared = true; }
    size_t na;
    NBE_TRY(read_counts(e, &na));
    const size_t total = (size_t)e->stats.tree_nodes;
    *count = total;
    const size_t m = total < cap ? total : cap;
    if (m == 0 || !out) return NBE_OK;
    if (e->force64) {
        std::vector<NodeD> h(m);
        NBE_HIP(hipMemcpyAsync(h.data(), e->tw.nodes, m * sizeof(NodeD), hipMemcpyDeviceToHost, e->stream));
        NBE_HIP(hipStreamSynchronize(e->stream));
        for (size_t i = 0; i < m; i++) {
            out[6 * i] = h[i].x; out[6 * i + 1] = h[i].y; out[6 * i + 2] = h[i].z; out[6 * i + 3] = h[i].m;
            out[6 * i + 4] = h[i].w; out[6 * i + 5] = (double)h[i].skip;
        }
    } else {
        std::vector<NodeF> h(m);
        NBE_HIP(hipMemcpyAsync(h.data(), e->tw.nodes, m * sizeof(NodeF), hipMemcpyDeviceToHost, e->stream));
        NBE_HIP(hipStreamSynchronize(e->stream));
        for (size_t i = 0; i < m; i++) {
            out[6 * i] = h[i].x; out[6 * i + 1] = h[i].y; out[6 * i + 2] = h[i].z; out[6 * i + 3] = h[i].m;
            // the stored threshold is (edge / theta)^2 + eps^2 for a cell, 0 for a leaf
            const double q = (double)h[i].w2 - (h[i].w2 > 0.0f ? e->cfg.eps * e->cfg.eps : 0.0);
            out[6 * i + 4] = sqrt(q > 0.0 ? q : 0.0) * e->cfg.theta; out[6 * i + 5] = (double)h[i].skip;
        }
    }
    return NBE_OK;
}

// Entry points that build host-side arrays of n elements: running out of host memory is reported, not thrown across
// the C boundary.
#define NBE_GUARDED(call)                                                                \
    try { return (call); }                                                               \
    catch (const std::bad_alloc&) { nbe_set_error("out of host memory"); return NBE_E_NOMEM; }

extern "C" int nbe_download(nbe_engine* e, double* pos, double* vel, double* mass, uint64_t* id, size_t* n_alive_out)
{
    if (e && multi_is_distributed(e)) { NBE_GUARDED(multi_dist_download(e, pos, vel, mass, id, n_alive_out)) }
    NBE_GUARDED(download_impl(NBE_FIRST(e), pos, vel, mass, id, n_alive_out))
}
extern "C" int nbe_set_radius(nbe_engine* e, const double* radius)
{
    if (e && e->multi) return multi_set_radius(e, radius);      // every GPU's engine holds the radii
    NBE_GUARDED(set_radius_impl(e, radius))
}
extern "C" int nbe_get_radius(nbe_engine* e, double* radius, size_t* n_alive_out)
{
    if (e && multi_is_distributed(e)) { NBE_GUARDED(multi_dist_get_radius(e, radius, n_alive_out)) }
    NBE_GUARDED(get_radius_impl(NBE_FIRST(e), radius, n_alive_out))
}
extern "C" int nbe_accel(nbe_engine* e, double* acc_xyz, uint8_t* alive_out)
{
    if (e && multi_is_distributed(e)) { NBE_GUARDED(multi_dist_accel(e, acc_xyz, alive_out)) }
    NBE_GUARDED(accel_impl(NBE_FIRST(e), acc_xyz, alive_out))
}
extern "C" int nbe_tree_nodes(nbe_engine* e, double* out, size_t cap, size_t* count)
{
    NBE_NOT_DISTRIBUTED(e, "nbe_tree_nodes");
    NBE_GUARDED(tree_nodes_impl(NBE_FIRST(e), out, cap, count))
}
