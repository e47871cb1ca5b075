/*
 * gkmkern_main.c -- standalone command-line front end of the gkm kernel-matrix path
 * (SURVEY.md §8(f3)).  Counterpart of the reference's debug CLI src/gkmkern_main.c:69-249
 * (`gkmkern posfile negfile outfile`, hard-coded L=10 k=6 d=3, 4 threads), with real options
 * and without its limits: the reference drops the last N mod 4 rows and overflows its
 * 10 000-double row buffers (gkmkern_main.c:58,187,221); this one writes every row.
 *
 *   gkmkern [-t type] [-l L] [-k k] [-d d] [-M M] [-H H] [-g gamma] [-T threads] [-v level] [-b]
 *           posfile negfile outfile
 *
 * Text output (default): row a = K(a,0..a-1) as "%e\t" followed by the literal "1.0", one
 * row per line -- the format of gkmkern_main.c:221-228.  -b: raw little-endian fp64 lower
 * triangle (row a: a+1 values) preceded by two int32 (n_pos, n_neg).
 * The matrix is computed through the same C ABI entry point the Python pipeline uses.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

#include "../../include/gkmkern_pylib.h"

static void usage(const char *prog)
{
    fprintf(stderr,
            "usage: %s [options] posfile negfile outfile\n"
            "  -t <0..5>  kernel type (default 4: wgkm)      -l <int>  word length L (default 11)\n"
            "  -k <int>   informative columns k (default 7)  -d <int>  max mismatches d (default 3)\n"
            "  -M <int>   weight decay start (default 50)    -H <num>  weight half life (default 50)\n"
            "  -g <num>   RBF gamma (default 1.0)            -T <int>  host threads for the copy-out (default 4)\n"
            "  -v <0..4>  verbosity (default 2)              -b        binary output\n",
            prog);
}

int main(int argc, char **argv)
{
    gkmOpt o;
    int binary = 0, c;
    memset(&o, 0, sizeof o);
    o.kernel_type = EST_TRUNC_PW; o.L = 11; o.k = 7; o.d = 3; o.M = 50; o.H = 50.0; o.gamma = 1.0;
    o.nthreads = 4; o.verbosity = 2;
    while ((c = getopt(argc, argv, "t:l:k:d:M:H:g:T:v:bh")) != -1) {
        switch (c) {
        case 't': o.kernel_type = atoi(optarg); break;
        case 'l': o.L = atoi(optarg); break;
        case 'k': o.k = atoi(optarg); break;
        case 'd': o.d = atoi(optarg); break;
        case 'M': o.M = (uint8_t)atoi(optarg); break;
        case 'H': o.H = atof(optarg); break;
        case 'g': o.gamma = atof(optarg); break;
        case 'T': o.nthreads = atoi(optarg); break;
        case 'v': o.verbosity = atoi(optarg); break;
        case 'b': binary = 1; break;
        default: usage(argv[0]); return c == 'h' ? 0 : 2;
        }
    }
    if (argc - optind != 3) { usage(argv[0]); return 2; }
    o.posfile = argv[optind];
    o.negfile = argv[optind + 1];
    const char *outfile = argv[optind + 2];

    gkm_problem *p = gkm_problem_read(o.posfile, o.negfile);
    if (!p) { fprintf(stderr, "cannot read %s / %s\n", o.posfile, o.negfile); return 1; }
    const int n = gkm_problem_size(p);
    gkm_problem_free(p);
    if (n <= 0) { fprintf(stderr, "no sequences\n"); return 1; }

    /* row a only needs a+1 doubles: a packed lower triangle, not an n x n square */
    double *tri = (double *)calloc((size_t)n * ((size_t)n + 1) / 2, sizeof(double));
    double **rows = (double **)malloc(sizeof(double *) * (size_t)n);
    if (!tri || !rows) { fprintf(stderr, "out of memory\n"); return 1; }
    for (int a = 0; a < n; a++) rows[a] = tri + (size_t)a * ((size_t)a + 1) / 2;
    int sizes[2] = {0, 0};
    if (gkm_main_pywrapper(&o, rows, sizes) != 0) { fprintf(stderr, "kernel computation failed\n"); return 1; }

    FILE *fo = fopen(outfile, binary ? "wb" : "w");
    if (!fo) { perror(outfile); return 1; }
    if (binary) {
        fwrite(sizes, sizeof(int), 2, fo);
        fwrite(tri, sizeof(double), (size_t)n * ((size_t)n + 1) / 2, fo);
    } else {
        for (int a = 0; a < n; a++) {
            for (int j = 0; j < a; j++) fprintf(fo, "%e\t", rows[a][j]);
            fprintf(fo, "1.0\n");
        }
    }
    fclose(fo);
    free(rows);
    free(tri);
    return 0;
}
