/* Plain C99 client: one engine handle driving several shards (nbe_config.num_devices), here all on device 0,
 * must give what an engine with a single shard gives: bit for bit in tree mode (main.go:134-149 fans the walk out
 * over contiguous body ranges against one shared tree; the ranges do not change any body's result). */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "nbody_engine.h"

#define N 5003   /* not a multiple of the shard count */

static double frand(unsigned *s) { *s = *s * 1664525u + 1013904223u; return (double)(*s >> 8) / 16777216.0; }

/* distributed: 0 = every shard holds all bodies; 1 = every body on one shard only (cfg.distributed; one shard: the
 * same cell sums on one engine, cfg.cell_sums = NBE_SUMS_EXACT, which is what a distributed run must reproduce) */
static int run_as(int mode, int precision, int shards, int distributed, double *pos_out, double *vel_out);
static int run(int mode, int precision, int shards, double *pos_out, double *vel_out) { return run_as(mode, precision, shards, 0, pos_out, vel_out); }

static int run_as(int mode, int precision, int shards, int distributed, double *pos_out, double *vel_out)
{
    static double pos[3 * N], vel[3 * N], mass[N];
    static const int ids[4] = {0, 0, 0, 0};
    nbe_config cfg;
    nbe_engine *e = NULL;
    nbe_stats_t st;
    size_t alive = 0;
    unsigned seed = 12345u;
    int i;
    for (i = 0; i < N; i++) {
        pos[3 * i] = 2.0e4 * (frand(&seed) - 0.5); pos[3 * i + 1] = 2.0e4 * (frand(&seed) - 0.5); pos[3 * i + 2] = 2.0e3 * (frand(&seed) - 0.5);
        vel[3 * i] = 0.01 * (frand(&seed) - 0.5); vel[3 * i + 1] = 0.01 * (frand(&seed) - 0.5); vel[3 * i + 2] = 0.0;
        mass[i] = 1.0e3 * (1.0 + frand(&seed));
    }
    pos[3 * 77] = 32767.9; vel[3 * 77] = 1.0;          /* leaves the bound in the first step (tree mode removes it) */
    if (nbe_default_config(&cfg) != NBE_OK) return 20;
    cfg.mode = mode; cfg.precision = precision; cfg.theta = 0.6;
    if (shards > 1) { cfg.num_devices = shards; cfg.device_ids = ids; cfg.distributed = distributed; }
    else if (distributed) cfg.cell_sums = NBE_SUMS_EXACT;
    if (nbe_create(&cfg, &e) != NBE_OK) { printf("create: %s\n", nbe_last_error()); return 21; }
    if (nbe_upload(e, N, pos, vel, mass, NULL) != NBE_OK) { printf("upload: %s\n", nbe_last_error()); return 22; }
    if (nbe_step(e, 7) != NBE_OK) { printf("step: %s\n", nbe_last_error()); return 23; }
    if (shards > 1 && nbe_step_timed(e, &st) != NBE_E_UNSUPPORTED) return 24;
    if (nbe_stats(e, &st) != NBE_OK || st.steps != 7) return 25;
    memset(pos_out, 0, sizeof(double) * 3 * N); memset(vel_out, 0, sizeof(double) * 3 * N);
    if (nbe_download(e, pos_out, vel_out, NULL, NULL, &alive) != NBE_OK) return 26;
    if (alive != (size_t)(mode == NBE_MODE_TREE ? N - 1 : N)) { printf("alive %zu\n", alive); return 27; }
    nbe_destroy(e);
    return 0;
}

int main(void)
{
    static double p1[3 * N], v1[3 * N], p2[3 * N], v2[3 * N];
    const int precisions[3] = {NBE_PREC_F32, NBE_PREC_MIXED, NBE_PREC_F64};
    int k, shards, rc, i;
    for (k = 0; k < 3; k++) {
        if ((rc = run(NBE_MODE_TREE, precisions[k], 1, p1, v1)) != 0) return rc;
        for (shards = 2; shards <= 4; shards++) {
            if ((rc = run(NBE_MODE_TREE, precisions[k], shards, p2, v2)) != 0) return rc;
            if (memcmp(p1, p2, sizeof(p1)) != 0 || memcmp(v1, v2, sizeof(v1)) != 0) { printf("tree precision %d, %d shards differ from one\n", precisions[k], shards); return 30; }
        }
    }
    for (k = 0; k < 3; k += 2) {       /* fp32 and fp64: the distributed tree equals one engine bit for bit */
        if ((rc = run_as(NBE_MODE_TREE, precisions[k], 1, 1, p1, v1)) != 0) return rc;
        for (shards = 2; shards <= 4; shards++) {
            if ((rc = run_as(NBE_MODE_TREE, precisions[k], shards, 1, p2, v2)) != 0) return rc;
            if (memcmp(p1, p2, sizeof(p1)) != 0 || memcmp(v1, v2, sizeof(v1)) != 0) { printf("distributed tree, precision %d, %d shards differ from one\n", precisions[k], shards); return 32; }
        }
    }
    if ((rc = run(NBE_MODE_DIRECT, NBE_PREC_F32, 1, p1, v1)) != 0) return rc;
    if ((rc = run(NBE_MODE_DIRECT, NBE_PREC_F32, 3, p2, v2)) != 0) return rc;
    for (i = 0; i < 3 * N; i++) {      /* a third of the targets per launch splits the source range differently: rounding only */
        const double d = p1[i] - p2[i];
        if (d > 1e-2 || d < -1e-2) { printf("direct: body %d differs by %g\n", i / 3, d); return 31; }
    }
    printf("abi multi ok\n");
    return 0;
}
