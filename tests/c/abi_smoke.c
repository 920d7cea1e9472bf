/* Plain C99 client of the engine's C ABI: what a cgo binding sees.  Builds without a GPU; `./abi_smoke run`
 * steps a tiny scene when a GPU is present, otherwise it checks that creation fails loudly. */
#include <stdio.h>
#include <string.h>
#include "nbody_engine.h"

int main(int argc, char **argv)
{
    nbe_config cfg;
    nbe_engine *e = NULL;
    double pos[6] = {0, 0, 0, 3, 4, 12}, vel[6] = {0, 0, 0, 0, 0, 0}, mass[2] = {3.0e6, 5.0e4}, acc[6];
    unsigned char alive[2];
    size_t n_alive = 0;
    int rc;
    (void)argv;
    if (nbe_default_config(&cfg) != NBE_OK) return 10;
    if (cfg.G != 6.67408e-11 || cfg.dt != 3600.0 || cfg.theta != 1.0 || cfg.bound_width != 65536.0 || cfg.collisions != 0) return 11;
    if (nbe_default_config(NULL) != NBE_E_INVALID || strstr(nbe_last_error(), "NULL") == NULL) return 12;
    rc = nbe_create(&cfg, &e);
    if (rc != NBE_OK) {
        printf("no engine: %s\n", nbe_last_error());
        return (rc == NBE_E_DEVICE && e == NULL && argc < 2) ? 0 : 13;   /* fine only when no run was asked for */
    }
    if (nbe_upload(e, 2, pos, vel, mass, NULL) != NBE_OK) return 14;
    if (nbe_accel(e, acc, alive) != NBE_OK) return 15;
    /* physics.go:135-145: a = G m / r^2 along the unit vector, r = 13 */
    {
        const double want = 6.67408e-11 * 5.0e4 / 169.0 * 3.0 / 13.0;
        if (acc[0] < want * (1 - 1e-5) || acc[0] > want * (1 + 1e-5) || !alive[0] || !alive[1]) return 16;
    }
    if (nbe_step(e, 3) != NBE_OK || nbe_download(e, pos, vel, mass, NULL, &n_alive) != NBE_OK || n_alive != 2) return 17;
    nbe_destroy(e);
    printf("abi smoke ok\n");
    return 0;
}
