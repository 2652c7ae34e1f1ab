/* A plain C99 caller of the C ABI (include/sigp.h): what a non-Python host of the GP block north/June1st.py:264-277 links against.
 * No Python, no torch, no C++ in this file; the library is the only dependency.
 *
 *   c_caller single  <n> <d> <m> <kernel 1|2> <dtype 0|1> <ell> <sn> <in.bin>
 *       one GPU: the fused call (sigp_fit_predict) and the step-by-step calls (kernel_build / potrf / fit / predict) on the same data.
 *   c_caller sharded <nranks> <n> <d> <m> <kernel 1|2> <dtype 0|1> <ell> <sn> <W> <in.bin>
 *       one process per rank, forked BEFORE anything touches the GPU; rank r uses device r % <visible devices> through the
 *       library's own RCCL communicator (sigp_dist_init); the 128-byte unique id travels from rank 0 to the others over pipes.
 *       RCCL wants one device per rank, so nranks > 1 needs that many GPUs; nranks = 1 runs every collective on a one-rank communicator.
 *
 * in.bin: doubles, X [n][d] then y [n] then Xs [m][d] (m >= 1 test points; they ride along the fit) then Xnew [m][d] (predicted after it).
 * Output (rank 0 / the single process), one value per line, "%.17g":
 *   sigma_f nlml info sigma_n | mean[m] | var[m] | mean_new[m] | var_new[m]
 * kernel: 1 = RBF, 2 = Matern-5/2 (SIGP_KERNEL_RBF / SIGP_KERNEL_MATERN52); dtype: 0 = fp64, 1 = fp32 factor + fp64 refinement. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <unistd.h>
#include <sys/types.h>
#include <sys/wait.h>

#include "sigp.h"

#define CHECK(h, call)                                                                      \
  do {                                                                                      \
    int rc_ = (call);                                                                       \
    if (rc_ != SIGP_OK) {                                                                   \
      fprintf(stderr, "%s -> %d: %s\n", #call, rc_, (h) ? sigp_last_error(h) : "(no handle)"); \
      return 10 + rc_;                                                                      \
    }                                                                                       \
  } while (0)

typedef struct {
  long n, d, m;
  int kernel, dtype;
  double ell, sn;
  double *X, *y, *Xs, *Xnew;
} problem;

static int read_problem(problem* p, const char* path) {
  const size_t nx = (size_t)(p->n * p->d), ns = (size_t)(p->m * p->d);
  const size_t total = nx + (size_t)p->n + 2 * ns;
  double* buf = (double*)malloc(total * sizeof(double));
  FILE* f = fopen(path, "rb");
  if (!buf || !f) { fprintf(stderr, "cannot open %s\n", path); return 1; }
  if (fread(buf, sizeof(double), total, f) != total) { fprintf(stderr, "%s: short file\n", path); fclose(f); return 1; }
  fclose(f);
  p->X = buf; p->y = buf + nx; p->Xs = p->y + p->n; p->Xnew = p->Xs + ns;
  return 0;
}

static void print_vec(const double* v, long k) {
  long i;
  for (i = 0; i < k; ++i) printf("%.17g\n", v[i]);
}

static int run_single(const problem* p) {
  sigp_handle* h = NULL;
  double out[4], sigma_f = 0, nlml = 0;
  int64_t info = 0;
  long i;
  double* r = (double*)calloc((size_t)(6 * p->m), sizeof(double));
  double *mean = r, *var = r + p->m, *mean_new = r + 2 * p->m, *var_new = r + 3 * p->m, *mean2 = r + 4 * p->m, *var2 = r + 5 * p->m;
  CHECK(h, sigp_create(&h, 0, p->dtype));
  CHECK(h, sigp_set_train(h, p->X, p->n, p->d, p->d, p->y));
  CHECK(h, sigp_set_test(h, p->Xs, p->m, p->d));
  /* the fused hot path: build -> Cholesky -> solves -> predictions at the ride points, one host synchronisation */
  CHECK(h, sigp_fit_predict(h, p->kernel, p->ell, p->sn, NULL, 0, out, mean, var));
  CHECK(h, sigp_predict(h, p->Xnew, p->m, p->d, mean_new, var_new));
  if (p->dtype == SIGP_F64) {
    /* the same statements one call at a time (north/June1st.py:265, :265, :266-271, :272-277); must agree bit for bit */
    CHECK(h, sigp_kernel_build(h, p->kernel, p->ell, p->sn));
    CHECK(h, sigp_potrf(h, &info));
    CHECK(h, sigp_fit(h, &sigma_f, &nlml));
    CHECK(h, sigp_predict_ride(h, mean2, var2));
    if (sigma_f != out[0] || nlml != out[1] || info != (int64_t)out[2]) { fprintf(stderr, "step-by-step path differs from the fused one\n"); return 3; }
    for (i = 0; i < p->m; ++i)
      if (mean2[i] != mean[i] || var2[i] != var[i]) { fprintf(stderr, "step-by-step predictions differ\n"); return 3; }
  }
  print_vec(out, 4);
  print_vec(mean, p->m); print_vec(var, p->m); print_vec(mean_new, p->m); print_vec(var_new, p->m);
  CHECK(h, sigp_destroy(h));
  free(r);
  return 0;
}

static int run_rank(const problem* p, int nranks, int rank, int ndev, long W, const int* id_rd, const int* id_wr) {
  sigp_handle* h = NULL;
  unsigned char id[128];
  double out[4], ranks_seen = 0;
  int k;
  double* r = (double*)calloc((size_t)(4 * p->m), sizeof(double));
  double *mean = r, *var = r + p->m, *mean_new = r + 2 * p->m, *var_new = r + 3 * p->m;
  CHECK(h, sigp_create(&h, rank % ndev, p->dtype));
  if (rank == 0) {
    CHECK(h, sigp_dist_unique_id(id));
    for (k = 1; k < nranks; ++k)
      if (write(id_wr[k], id, sizeof id) != (ssize_t)sizeof id) { perror("write id"); return 4; }
  } else if (read(id_rd[rank], id, sizeof id) != (ssize_t)sizeof id) { perror("read id"); return 4; }
  CHECK(h, sigp_set_option(h, "owner_only", 1));          /* no n x n matrix on any rank: each stores its own block columns only */
  CHECK(h, sigp_dist_init(h, nranks, rank, id));
  CHECK(h, sigp_set_train(h, p->X, p->n, p->d, p->d, p->y));
  CHECK(h, sigp_set_test(h, p->Xs, p->m, p->d));
  CHECK(h, sigp_dist_fit(h, p->kernel, p->ell, p->sn, NULL, 0, W, 1, out, mean, var));
  CHECK(h, sigp_dist_predict(h, p->Xnew, p->m, p->d, mean_new, var_new));
  CHECK(h, sigp_get_stat(h, "dist_comm_ranks", &ranks_seen));
  if ((int)ranks_seen != nranks) { fprintf(stderr, "rank %d: the communicator reports %d ranks, expected %d\n", rank, (int)ranks_seen, nranks); return 5; }
  if (rank == 0) {
    print_vec(out, 4);
    print_vec(mean, p->m); print_vec(var, p->m); print_vec(mean_new, p->m); print_vec(var_new, p->m);
  }
  CHECK(h, sigp_dist_shutdown(h));
  CHECK(h, sigp_destroy(h));
  free(r);
  return 0;
}

int main(int argc, char** argv) {
  problem p;
  int a = 2;
  memset(&p, 0, sizeof p);
  if (argc >= 10 && !strcmp(argv[1], "single")) {
    p.n = atol(argv[a]); p.d = atol(argv[a + 1]); p.m = atol(argv[a + 2]); p.kernel = atoi(argv[a + 3]); p.dtype = atoi(argv[a + 4]);
    p.ell = atof(argv[a + 5]); p.sn = atof(argv[a + 6]);
    if (p.m < 1 || p.m > SIGP_MAX_RIDE || read_problem(&p, argv[a + 7])) return 2;
    return run_single(&p);
  }
  if (argc >= 12 && !strcmp(argv[1], "sharded")) {
    const int nranks = atoi(argv[2]);
    int ndev = 1, rank, status = 0, worst = 0;
    long W;
    int (*pipes)[2];
    int *rd, *wr;
    pid_t* pids;
    const char* env = getenv("SIGP_EXAMPLE_DEVICES");   /* visible devices (ranks map onto them round-robin); default = nranks */
    a = 3;
    p.n = atol(argv[a]); p.d = atol(argv[a + 1]); p.m = atol(argv[a + 2]); p.kernel = atoi(argv[a + 3]); p.dtype = atoi(argv[a + 4]);
    p.ell = atof(argv[a + 5]); p.sn = atof(argv[a + 6]); W = atol(argv[a + 7]);
    if (nranks < 1 || nranks > 64 || p.m < 1 || p.m > SIGP_MAX_RIDE || read_problem(&p, argv[a + 8])) return 2;
    ndev = env ? atoi(env) : nranks;
    if (ndev < 1) ndev = 1;
    pipes = (int (*)[2])calloc((size_t)nranks, sizeof *pipes);
    rd = (int*)calloc((size_t)nranks, sizeof(int)); wr = (int*)calloc((size_t)nranks, sizeof(int));
    pids = (pid_t*)calloc((size_t)nranks, sizeof(pid_t));
    for (rank = 1; rank < nranks; ++rank) {
      if (pipe(pipes[rank])) { perror("pipe"); return 4; }
      rd[rank] = pipes[rank][0]; wr[rank] = pipes[rank][1];
    }
    /* fork first, touch the GPU afterwards: a HIP context does not survive fork() */
    for (rank = 1; rank < nranks; ++rank) {
      pids[rank] = fork();
      if (pids[rank] < 0) { perror("fork"); return 4; }
      if (pids[rank] == 0) _exit(run_rank(&p, nranks, rank, ndev, W, rd, wr));
    }
    worst = run_rank(&p, nranks, 0, ndev, W, rd, wr);
    for (rank = 1; rank < nranks; ++rank) {
      if (waitpid(pids[rank], &status, 0) < 0 || !WIFEXITED(status) || WEXITSTATUS(status) != 0) {
        fprintf(stderr, "rank %d failed (status %d)\n", rank, status);
        if (!worst) worst = 6;
      }
    }
    return worst;
  }
  fprintf(stderr, "usage: %s single <n> <d> <m> <kernel> <dtype> <ell> <sn> <in.bin>\n       %s sharded <nranks> <n> <d> <m> <kernel> <dtype> <ell> <sn> <W> <in.bin>\n", argv[0], argv[0]);
  return 2;
}
