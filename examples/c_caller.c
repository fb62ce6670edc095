/* A plain C99 caller of the C ABI (include/tcsfm.h): no torch, no C++, host arrays in, host arrays out.
 *
 *   gcc -std=c99 -O2 -Iinclude examples/c_caller.c -Ltightly_coupled_sfm_amd -ltcsfm_hip -Wl,-rpath,$PWD/tightly_coupled_sfm_amd -lm -o c_caller
 *   ./c_caller            (needs an MI355X; prints the refined poses of two synthetic directed pairs)
 *
 * The two "frames" are a smooth procedural texture seen through a small sideways shift, with a constant depth plane --
 * enough for the refinement to have something to do; the parity tests use the proper synthetic scenes. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include "tcsfm.h"

#define H 96
#define W 320

static float tex(float x, float y, int c) { return 0.5f + 0.25f * sinf(0.11f * x + 0.7f * c) * cosf(0.07f * y - 0.3f * c) + 0.2f * sinf(0.031f * (x + y)); }

int main(void) {
    const int N = 2, hw = H * W;
    float *tgt = malloc(sizeof(float) * N * 3 * hw), *src = malloc(sizeof(float) * N * 3 * hw);
    float *dt = malloc(sizeof(float) * N * hw), *ds = malloc(sizeof(float) * N * hw);
    float K[2 * 9], pose_in[2 * 6] = {0}, pose_out[2 * 6], stats[2 * 5 * TCSFM_NSTAT];
    if (!tgt || !src || !dt || !ds) return 2;
    for (int n = 0; n < N; n++) {
        const float shift = n == 0 ? 1.5f : -1.5f;   /* pixels of horizontal parallax between the two frames */
        for (int c = 0; c < 3; c++)
            for (int v = 0; v < H; v++)
                for (int u = 0; u < W; u++) {
                    tgt[((n * 3 + c) * H + v) * W + u] = tex((float)u, (float)v, c);
                    src[((n * 3 + c) * H + v) * W + u] = tex((float)u + shift, (float)v, c);
                }
        for (int i = 0; i < hw; i++) dt[n * hw + i] = ds[n * hw + i] = 1.0f;
        const float k[9] = {184.6f, 0, 157.2f, 0, 183.5f, 47.5f, 0, 0, 1};
        for (int i = 0; i < 9; i++) K[n * 9 + i] = k[i];
    }
    tcsfm_handle h = NULL;
    int rc = tcsfm_create(&h, 0, H, W, N);
    if (rc != TCSFM_OK) { fprintf(stderr, "tcsfm_create: %d %s\n", rc, tcsfm_last_error(NULL)); return 1; }
    tcsfm_opts o;
    tcsfm_default_opts(&o);
    o.host_ptrs = 1;          /* every array below lives in host memory; the library stages it */
    o.n_iters = 4;
    rc = tcsfm_refine(h, &o, N, tgt, src, dt, ds, K, pose_in, NULL, pose_out, NULL, stats);
    if (rc != TCSFM_OK) { fprintf(stderr, "tcsfm_refine: %d %s\n", rc, tcsfm_last_error(h)); tcsfm_destroy(h); return 1; }
    for (int n = 0; n < N; n++) {
        printf("pair %d: cost %.5f -> %.5f   pose [", n, stats[(n * 5 + 0) * TCSFM_NSTAT], stats[(n * 5 + 3) * TCSFM_NSTAT]);
        for (int i = 0; i < 6; i++) printf("%s% .5f", i ? ", " : "", pose_out[n * 6 + i]);
        printf("]\n");
    }
    tcsfm_destroy(h);
    free(tgt); free(src); free(dt); free(ds);
    return 0;
}
