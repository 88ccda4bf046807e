/* Plain-C caller of the C ABI (include/boundmpc_hip.h): no Python, no torch.  Reads one problem (p, x0) from a binary file of
 * doubles [505 | 440], solves it with bmpc_solve_batch_host and prints "status iters f x[8..14]".  Built and run by
 * tests/test_gpu_parity.py::test_plain_c_caller_of_the_abi on the GPU box (gcc + the in-tree libboundmpc_hip.so). */
#include <stdio.h>
#include <stdlib.h>
#include "../../include/boundmpc_hip.h"

int main(int argc, char **argv) {
    if (argc < 2) { fprintf(stderr, "usage: %s problem.bin\n", argv[0]); return 2; }
    enum { NP = 505, NW = 440, NG = 430 };
    static double p[NP], x0[NW], x[NW], g[NG], lam_g[NG], lam_x[NW], f, kkt;
    int iters = -1, status = -1;
    FILE *fh = fopen(argv[1], "rb");
    if (!fh || fread(p, sizeof(double), NP, fh) != NP || fread(x0, sizeof(double), NW, fh) != NW) { fprintf(stderr, "cannot read %s\n", argv[1]); return 2; }
    fclose(fh);
    bmpc_options o;
    bmpc_default_options(&o);
    bmpc_handle *h = NULL;
    int rc = bmpc_create(10, 4, 0.1, &o, &h);
    if (rc != BMPC_OK) { fprintf(stderr, "bmpc_create: %s\n", bmpc_error_string(rc)); return 1; }
    if (bmpc_num_vars(h) != NW || bmpc_num_cons(h) != NG || bmpc_num_params(h) != NP) { fprintf(stderr, "size mismatch\n"); return 1; }
    rc = bmpc_solve_batch_host(h, 1, p, x0, x, g, lam_g, lam_x, &f, &iters, &status, &kkt);
    if (rc != BMPC_OK) { fprintf(stderr, "bmpc_solve_batch_host: %s\n", bmpc_error_string(rc)); return 1; }
    printf("%d %d %.17g", status, iters, f);
    for (int i = 8; i < 15; i++) printf(" %.17g", x[i]);
    printf("\n");
    bmpc_destroy(h);
    return 0;
}
