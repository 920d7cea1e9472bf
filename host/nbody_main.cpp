// nbody -- C++ host of the GPU gravity engine, mirroring the reference's driver (main.go) above the C ABI.
//
// The reference is a Go program; no Go toolchain exists where this was written, so the host side is C++ with the
// reference's flags, its body type, its initial-condition generator and its frame loop:
//   flags           main.go:17-24   -n -y -state -save -tree -norender -nocollision
//                   (+ -seed -frames-dir -precision -asgo -perwave, and the reference's compile-time constants as
//                   run-time values: -gpus N (one engine, N GPUs; -devices 0,0,1 names them, an ordinal may repeat), -theta main.go:73, -dt main.go:28, -bound main.go:74-76, -eps softening, -G physics.go:132)
//   makebodies      physics.go:16-73 with the two cores of main.go:35-36
//   frame loop      main.go:80-190  (frames+1 iterations; a frame copy goes to the workers BEFORE each step)
//   output workers  main.go:50-61   4 consumers behind a queue of capacity 32 (here fed by nbe_frame_stream_begin)
//   save / resume   main.go:199-248 (the reference's gob stream of a frameJob, host/gob.hpp; -recode FILE reads a
//                   state file and writes it back as %010d.data, without touching the GPU)
// The step itself (main.go:104-178) is one nbe_step call, collisions included: in tree mode the search, queue and
// merge of tree.go:245-281 / main.go:151-167, in the direct sum the merge inside the pair sweep (main.go:122-126).
// The reference's sinks are a PNG renderer, a gob chunk store and sqlite; the sink here writes the record the
// gob store keeps per body (gob.go:54-61: X, Y, Z, Mass, Radius as float32, keyed by ID) as raw binary frames;
// `python -m nbody_amd.render DIR` turns those into the images of render.go.
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <filesystem>
#include <memory>
#include <mutex>
#include <random>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

#include "gob.hpp"
#include "nbody_engine.h"

// physics.go:91-98
struct body {
    uint64_t ID = 0;
    double Mass = 0, Radius = 0;
    double X = 0, Y = 0, Z = 0;
    double Vx = 0, Vy = 0, Vz = 0;
    double fx = 0, fy = 0, fz = 0;
};

// render.go:33-36
struct frameJob {
    int Frame = 0;
    std::vector<body> Bodies;
};

static const double G = 6.67408e-11;   // physics.go:132

[[noreturn]] static void panic(const std::string& msg)   // the reference panics on every error (main.go:213-238)
{
    std::fprintf(stderr, "panic: %s\n", msg.c_str());
    std::exit(2);
}
static void check(int rc) { if (rc != 0) panic(nbe_last_error()); }

// physics.go:76-81
static void cross(double x1, double y1, double z1, double x2, double y2, double z2, double& x3, double& y3, double& z3)
{
    x3 = y1 * z2 - z1 * y2;
    y3 = z1 * x2 - x1 * z2;
    z3 = x1 * y2 - y1 * x2;
}

// physics.go:16-73.  The reference draws from Go's global math/rand stream; here a seeded mt19937_64.
static std::vector<body> makebodies(int n, std::vector<body> cores, uint64_t seed)
{
    const double orbitalVDampening = 1.0, meanMass = 1e3, defaultRadius = 2, galaxyRadialStdDev = 1e3, galaxyAxialStdDev = 1e1;
    std::mt19937_64 rng(seed);
    std::normal_distribution<double> norm(0.0, 1.0);
    const int nc = (int)cores.size();
    std::vector<body> bodies((size_t)(n + nc));
    for (int i = nc; i < n + nc; i++) {
        const double m = std::fabs(norm(rng) * 500 + meanMass);
        body core;
        if (nc > 0) core = cores[(size_t)(rng() % (uint64_t)nc)];
        body& b = bodies[(size_t)i];
        b.ID = (uint64_t)i;
        b.Mass = m;
        b.Radius = defaultRadius;
        b.X = norm(rng) * (galaxyRadialStdDev * (1 - std::fabs(core.fx)) + galaxyAxialStdDev * std::fabs(core.fx)) + core.X;
        b.Y = norm(rng) * (galaxyRadialStdDev * (1 - std::fabs(core.fy)) + galaxyAxialStdDev * std::fabs(core.fy)) + core.Y;
        b.Z = norm(rng) * (galaxyRadialStdDev * (1 - std::fabs(core.fz)) + galaxyAxialStdDev * std::fabs(core.fz)) + core.Z;
        if (nc > 0) {
            double dx = core.X - b.X, dy = core.Y - b.Y, dz = core.Z - b.Z;
            double d = std::sqrt(dx * dx + dy * dy + dz * dz);
            if (d == 0) d = 1;
            dx /= d; dy /= d; dz /= d;
            cross(dx, dy, dz, core.fx, core.fy, core.fz, dx, dy, dz);
            const double v = std::sqrt(G * core.Mass / d);
            b.Vx = dx * v * orbitalVDampening + core.Vx;
            b.Vy = dy * v * orbitalVDampening + core.Vy;
            b.Vz = dz * v * orbitalVDampening + core.Vz;
        }
    }
    for (int i = 0; i < nc; i++) {   // cores go to the front with their rotation axis cleared (physics.go:66-70)
        cores[(size_t)i].fx = cores[(size_t)i].fy = cores[(size_t)i].fz = 0;
        bodies[(size_t)i] = cores[(size_t)i];
        bodies[(size_t)i].ID = (uint64_t)i;
    }
    return bodies;
}

// ---- save / resume (main.go:210-248): the reference's own file, a gob stream of the frameJob ----
static void exportFrameData(const frameJob& frame)
{
    char fname[64];
    std::snprintf(fname, sizeof(fname), "%010d.data", frame.Frame);      // main.go:211
    gob::FrameJob job;
    job.Frame = frame.Frame;
    job.Bodies.reserve(frame.Bodies.size());
    for (const body& b : frame.Bodies) {   // exported fields only (physics.go:92-96); the force accumulators do not travel
        gob::Body g;
        g.ID = b.ID; g.Mass = b.Mass; g.Radius = b.Radius; g.X = b.X; g.Y = b.Y; g.Z = b.Z; g.Vx = b.Vx; g.Vy = b.Vy; g.Vz = b.Vz;
        job.Bodies.push_back(g);
    }
    const std::string bytes = gob::encode(job);
    FILE* f = std::fopen(fname, "wb");
    if (!f) panic(std::string("cannot create ") + fname);
    const bool ok = std::fwrite(bytes.data(), 1, bytes.size(), f) == bytes.size();
    std::fclose(f);
    if (!ok) { std::remove(fname); panic(std::string("write failed: ") + fname); }   // main.go:220-223
}

static frameJob importFrameData(const std::string& filename)
{
    FILE* f = std::fopen(filename.c_str(), "rb");
    if (!f) panic("cannot open " + filename);
    std::string bytes;
    char buf[1 << 16];
    for (size_t k; (k = std::fread(buf, 1, sizeof(buf), f)) > 0;) bytes.append(buf, k);
    std::fclose(f);
    gob::FrameJob g;
    try {
        g = gob::decode(bytes.data(), bytes.size());
    } catch (const std::exception& ex) {
        panic(filename + ": " + ex.what());                                // main.go:236-238
    }
    frameJob job;
    job.Frame = (int)g.Frame;
    job.Bodies.resize(g.Bodies.size());
    for (size_t k = 0; k < g.Bodies.size(); k++) {
        body& b = job.Bodies[k];
        const gob::Body& s = g.Bodies[k];
        b.ID = s.ID; b.Mass = s.Mass; b.Radius = s.Radius; b.X = s.X; b.Y = s.Y; b.Z = s.Z; b.Vx = s.Vx; b.Vy = s.Vy; b.Vz = s.Vz;
    }
    return job;
}

// ---- output workers: 4 consumers behind a bounded queue, the shape of main.go:50-61 ----
struct FrameQueue {
    std::mutex mu;
    std::condition_variable cv_put, cv_get;
    std::deque<std::unique_ptr<frameJob>> q;
    size_t cap = 32;                // main.go:50
    bool closed = false;
    void put(std::unique_ptr<frameJob> j)
    {
        std::unique_lock<std::mutex> lk(mu);
        cv_put.wait(lk, [&] { return q.size() < cap; });
        q.push_back(std::move(j));
        cv_get.notify_one();
    }
    std::unique_ptr<frameJob> get()
    {
        std::unique_lock<std::mutex> lk(mu);
        cv_get.wait(lk, [&] { return closed || !q.empty(); });
        if (q.empty()) return nullptr;
        auto j = std::move(q.front());
        q.pop_front();
        cv_put.notify_one();
        return j;
    }
    void close()
    {
        std::lock_guard<std::mutex> lk(mu);
        closed = true;
        cv_get.notify_all();
    }
};

// Sink with the role of frameToImages / frameToMemory (render.go:97, gob.go:188): one file per frame holding
// n, then per body its ID (u32) and the renderbody record of gob.go:54-61.
static void frameToFiles(const std::string& dir, FrameQueue* ch)
{
    while (auto job = ch->get()) {
        char name[64];
        std::snprintf(name, sizeof(name), "/%010d.frame", job->Frame);
        FILE* f = std::fopen((dir + name).c_str(), "wb");
        if (!f) panic("cannot create " + dir + name);
        const uint32_t n = (uint32_t)job->Bodies.size();
        std::fwrite(&n, 4, 1, f);
        for (const body& b : job->Bodies) {
            const uint32_t id = (uint32_t)b.ID;
            const float rec[5] = {(float)b.X, (float)b.Y, (float)b.Z, (float)b.Mass, (float)b.Radius};
            std::fwrite(&id, 4, 1, f);
            std::fwrite(rec, sizeof(rec), 1, f);
        }
        std::fclose(f);
    }
}

struct StreamCtx {
    FrameQueue* ch;
    const std::unordered_map<uint64_t, double>* radius_by_id;   // used when the engine sends no radii (collisions off: they never change)
    std::mutex mu;
    std::unique_ptr<frameJob> last;            // lastFrame of main.go:97-100
    int remaining = 0;
};

// Runs on the engine's delivery thread: turns the streamed arrays into a frameJob and sends it on (main.go:96-101).
static void on_frame(void* user, int frame, size_t n, const float* xyzm, const float* radius, const uint64_t* ids)
{
    StreamCtx* ctx = (StreamCtx*)user;
    auto job = std::make_unique<frameJob>();
    job->Frame = frame;
    job->Bodies.resize(n);
    for (size_t k = 0; k < n; k++) {
        body& b = job->Bodies[k];
        b.ID = ids[k];
        b.X = xyzm[4 * k]; b.Y = xyzm[4 * k + 1]; b.Z = xyzm[4 * k + 2]; b.Mass = xyzm[4 * k + 3];
        if (radius) b.Radius = radius[k];
        else {
            const auto it = ctx->radius_by_id->find(ids[k]);
            b.Radius = it != ctx->radius_by_id->end() ? it->second : 2.0;
        }
    }
    {
        std::lock_guard<std::mutex> lk(ctx->mu);
        ctx->remaining = (int)n;
    }
    ctx->ch->put(std::move(job));
}

int main(int argc, char** argv)
{
    // flags, main.go:17-25 (Go's single-dash long flags)
    int numbodies = 10;
    double years = 1;
    std::string stateFilename, framesDir = "frames";
    bool stateSave = false, tree = false, norender = false, nocollision = false, asgo = false, perwave = false, distributed = false;
    uint64_t seed = 1;
    double theta = 1;                                        // main.go:73
    double dt = 60 * 60;                                     // main.go:28
    double eps = 0, bound = 65536, gconst = G;               // no softening; simbound main.go:74-76; physics.go:132
    int precision = NBE_PREC_F32;
    int gpus = 1;                                            // the walk's fan-out (main.go:134-149) over this many GPUs
    std::vector<int> devices;                                // -devices: their ordinals (default 0 .. gpus-1)
    for (int i = 1; i < argc; i++) {
        std::string a = argv[i];
        if (a.rfind("--", 0) == 0) a = a.substr(1);
        auto val = [&](const char* name) -> const char* {
            const std::string pre = std::string(name) + "=";
            if (a.rfind(pre, 0) == 0) return argv[i] + (argv[i][1] == '-' ? 1 : 0) + pre.size();
            if (a == name && i + 1 < argc) return argv[++i];
            return nullptr;
        };
        if (const char* v = val("-n")) numbodies = std::atoi(v);
        else if (const char* v = val("-y")) years = std::atof(v);
        else if (const char* v = val("-state")) stateFilename = v;
        else if (const char* v = val("-seed")) seed = std::strtoull(v, nullptr, 10);
        else if (const char* v = val("-theta")) theta = std::atof(v);
        else if (const char* v = val("-dt")) dt = std::atof(v);
        else if (const char* v = val("-gpus")) gpus = std::atoi(v);
        else if (const char* v = val("-devices")) {
            devices.clear();
            for (const char* p = v; *p;) {
                char* end = nullptr;
                devices.push_back((int)std::strtol(p, &end, 10));
                if (end == p) panic(std::string("-devices: not a list of ordinals: ") + v);
                p = *end == ',' ? end + 1 : end;
            }
            gpus = (int)devices.size();
        }
        else if (const char* v = val("-eps")) eps = std::atof(v);
        else if (const char* v = val("-bound")) bound = std::atof(v);
        else if (const char* v = val("-G")) gconst = std::atof(v);
        else if (const char* v = val("-frames-dir")) framesDir = v;
        else if (const char* v = val("-precision")) precision = !std::strcmp(v, "f64") ? NBE_PREC_F64 : !std::strcmp(v, "mixed") ? NBE_PREC_MIXED : NBE_PREC_F32;
        else if (const char* v = val("-recode")) { exportFrameData(importFrameData(v)); return 0; }
        else if (a == "-save") stateSave = true;
        else if (a == "-distributed") distributed = true;
        else if (a == "-tree") tree = true;
        else if (a == "-norender") norender = true;
        else if (a == "-nocollision") nocollision = true;
        else if (a == "-asgo") asgo = true;
        else if (a == "-perwave") perwave = true;
        else panic("flag provided but not defined: " + a);
    }

    const int iterPerFrame = 1;                              // main.go:29
    int frames = (int)(years * 365 * 24) / iterPerFrame;     // main.go:30
    body c0, c1;                                             // main.go:35-36
    c0.Mass = 1e10; c0.Radius = 1.0; c0.X = -9000; c0.Y = -100; c0.Z = -2000; c0.fy = 1; c0.Vx = 0.004; c0.Vz = -0.001;
    c1.Mass = 1e10; c1.Radius = 1.0; c1.X = 9000; c1.Y = 100; c1.Z = 2000; c1.fz = -1; c1.Vx = -0.003; c1.Vz = 0.002;
    std::vector<body> bodies = makebodies(numbodies, {c0, c1}, seed);

    auto lastFrame = std::make_unique<frameJob>();
    int startFrame = 0;
    if (!stateFilename.empty()) {                            // main.go:42-47
        frameJob st = importFrameData(stateFilename);
        bodies = st.Bodies;                                  // overwriteBodies, main.go:243-248
        startFrame = st.Frame;
        frames += startFrame;
        *lastFrame = std::move(st);
    }

    FrameQueue ch;
    const int workers = 4;                                   // main.go:51
    std::vector<std::thread> wg;
    if (!norender) {
        std::error_code ec;
        std::filesystem::create_directories(framesDir, ec);
        if (ec) panic("cannot create " + framesDir + ": " + ec.message());
        for (int i = 0; i < workers; i++) wg.emplace_back(frameToFiles, framesDir, &ch);
    }

    std::printf("render: %s\ncollisions: %s\ntree: %s\nbodies: %zu\nstep: %g sec\nframes: %d\nsimulation time: %.1f days\n",
                norender ? "false" : "true", nocollision ? "false" : "true", tree ? "true" : "false", bodies.size(), dt,
                frames - startFrame, (double)dt * iterPerFrame * (frames - startFrame) / 3600.0 / 24.0);   // main.go:64-71

    nbe_config cfg;
    check(nbe_default_config(&cfg));                         // simbound and G as main.go:74-76, physics.go:132
    cfg.mode = tree ? NBE_MODE_TREE : NBE_MODE_DIRECT;
    cfg.theta = theta;
    cfg.dt = dt;
    cfg.eps = eps;
    cfg.bound_width = bound;
    cfg.G = gconst;
    cfg.precision = precision;
    cfg.collisions = nocollision ? 0 : 1;                    // main.go:24
    cfg.num_devices = gpus;                                  // > 1: devices 0 .. gpus-1 (with collisions on: -tree only)
    if (!devices.empty()) cfg.device_ids = devices.data();
    // -distributed (with -gpus N -tree): every body on one GPU only, pruned subtrees exchanged
    cfg.distributed = distributed ? 1 : 0;
    // -asgo: the tree path with the Go program's own arithmetic (running cell centres tree.go:189-193, last
    // n mod 4 bodies unwalked main.go:134-148), for comparing a run with the reference number for number
    if (asgo && tree) cfg.reference_quirks = NBE_QUIRK_REF_COM | NBE_QUIRK_DROP_TAIL;
    // -perwave: the bodies of a wavefront open cells together (faster, finer than tree.go:216-217 asks for, not the
    // reference's interaction set)
    if (perwave && tree && !asgo) cfg.walk_criterion = NBE_WALK_PER_WAVE;
    nbe_engine* eng = nullptr;
    check(nbe_create(&cfg, &eng));
    const size_t n = bodies.size();
    std::vector<double> pos(3 * n), vel(3 * n), mass(n);
    std::unordered_map<uint64_t, double> radius_by_id;     // IDs of a -state file are arbitrary 64-bit values
    radius_by_id.reserve(n);
    std::vector<uint64_t> id(n);
    for (size_t k = 0; k < n; k++) {
        const body& b = bodies[k];
        pos[3 * k] = b.X; pos[3 * k + 1] = b.Y; pos[3 * k + 2] = b.Z;
        vel[3 * k] = b.Vx; vel[3 * k + 1] = b.Vy; vel[3 * k + 2] = b.Vz;
        mass[k] = b.Mass; id[k] = b.ID;
        radius_by_id[b.ID] = b.Radius;
    }
    check(nbe_upload(eng, n, pos.data(), vel.data(), mass.data(), id.data()));
    std::vector<double> radius(n);
    for (size_t k = 0; k < n; k++) radius[k] = bodies[k].Radius;
    if (cfg.collisions) check(nbe_set_radius(eng, radius.data()));

    StreamCtx ctx;
    ctx.ch = &ch;
    ctx.radius_by_id = &radius_by_id;
    ctx.remaining = (int)n;
    if (!norender) check(nbe_frame_stream_begin(eng, iterPerFrame, 32, startFrame, on_frame, &ctx));

    const auto start = std::chrono::steady_clock::now();
    for (int frame = startFrame; frame <= frames; frame++) {          // main.go:80: frames+1 iterations
        check(nbe_step(eng, iterPerFrame));                           // main.go:104-178 (frame copy streams out first)
        int remainingBodies;
        if (norender) {                                               // main.go:83-88
            nbe_stats_t st;
            check(nbe_stats(eng, &st));
            remainingBodies = (int)st.n_alive;
        } else {
            std::lock_guard<std::mutex> lk(ctx.mu);
            remainingBodies = ctx.remaining;
        }
        if ((frame - startFrame) % 16 == 0 || frame == frames) {      // progress line, main.go:181-189
            const double elapsed = std::chrono::duration<double>(std::chrono::steady_clock::now() - start).count();
            const double avg = 1e3 * elapsed / (frame - startFrame + 1);
            std::printf("%.1f%%, %d bodies, %.0fms/frame, %.0fs remaining, %.0fs elapsed                    \r",
                        frames > startFrame ? 100.0 * (frame - startFrame) / (frames - startFrame) : 100.0, remainingBodies, avg,
                        avg * (frames - frame) / 1e3, elapsed);
            std::fflush(stdout);
        }
    }
    if (!norender) check(nbe_frame_stream_end(eng));
    ch.close();                                                       // main.go:191-193
    for (std::thread& t : wg) t.join();
    check(nbe_sync(eng));
    std::printf("\nDone. Took %.0fs\n", std::chrono::duration<double>(std::chrono::steady_clock::now() - start).count());

    if (stateSave) {   // main.go:199-201.  The reference saves lastFrame = the state BEFORE the final step (and only
                       // when rendering refreshed it, main.go:97-100); this host saves the final state, which is
                       // what a resumed run needs.
        size_t alive = 0;
        check(nbe_download(eng, pos.data(), vel.data(), mass.data(), id.data(), &alive));
        if (cfg.collisions) check(nbe_get_radius(eng, radius.data(), nullptr));
        frameJob fin;
        fin.Frame = frames + 1;
        fin.Bodies.resize(alive);
        for (size_t k = 0; k < alive; k++) {
            body& b = fin.Bodies[k];
            b.ID = id[k]; b.Mass = mass[k];
            const auto it = radius_by_id.find(id[k]);
            b.Radius = cfg.collisions ? radius[k] : it != radius_by_id.end() ? it->second : 2.0;
            b.X = pos[3 * k]; b.Y = pos[3 * k + 1]; b.Z = pos[3 * k + 2];
            b.Vx = vel[3 * k]; b.Vy = vel[3 * k + 1]; b.Vz = vel[3 * k + 2];
        }
        exportFrameData(fin);
    }
    nbe_destroy(eng);
    return 0;
}
