/* abi_smoke.c — plain-C caller of include/rsf_abi.h (test-only).  Mirrors INTEGRATION.md section 3:
 * one forward solve + SSq for three Dc values, then a short fused MCMC run; prints the numbers so the
 * Python test can compare them with the same calls made through ctypes.
 *   gcc abi_smoke.c -I../../include -L<csrc> -lrsf_hip -L/opt/rocm/lib -lamdhip64 -lm */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include "rsf_abi.h"

#define CHECK(x) do { if ((x) != RSF_OK) { fprintf(stderr, "%s failed: %s\n", #x, rsf_last_error()); return 1; } } while (0)

int main(void) {
  rsf_ctx *ctx;
  rsf_config cfg = {sizeof cfg, RSF_ABI_VERSION, -1, RSF_MEM_HOST, NULL, 0, 0};
  CHECK(rsf_create(&cfg, &ctx));
  rsf_model m = {sizeof m, RSF_FLAG_RADIATION_DAMPING, 500, 1, 0.0, 50.0, 0.6, 1.0, 1e-7, 0.6, 0.011, 0.014};
  CHECK(rsf_set_model(ctx, &m));
  int32_t nout;
  CHECK(rsf_model_nout(ctx, &nout));
  double dc[3] = {500.0, 1000.0, 2000.0}, ssq[3];
  double *acc = malloc(sizeof(double) * nout * 3), *data = malloc(sizeof(double) * nout);
  CHECK(rsf_forward_batch(ctx, 3, dc, NULL, NULL, NULL, NULL, acc));
  for (int k = 0; k < nout; ++k) data[k] = acc[k * 3 + 1] * (1.0 + 0.3 * sin(0.7 * k)); /* deterministic "noise" */
  CHECK(rsf_forward_batch(ctx, 3, dc, NULL, NULL, data, ssq, NULL));
  printf("nout %d\nssq %.17g %.17g %.17g\n", nout, ssq[0], ssq[1], ssq[2]);
  enum { C = 64, ITERS = 10 };
  double q0[C], tq[ITERS * C], ts[ITERS * C];
  for (int i = 0; i < C; ++i) q0[i] = 1000.0;
  rsf_mcmc_config mc = {sizeof mc, 1, C, 0, 2025, 0.01, 3, RSF_ADAPT_NONE, 10, 0, 1e-6, {0.0}, {1e4}};
  CHECK(rsf_mcmc_init(ctx, &mc, q0, data));
  CHECK(rsf_mcmc_run(ctx, ITERS, tq, ts, NULL));
  int64_t acc_n, eval_n, nonfinite, done;
  CHECK(rsf_mcmc_stats(ctx, &acc_n, &eval_n, &nonfinite, &done));
  double mean = 0;
  for (int i = 0; i < C; ++i) mean += tq[(ITERS - 1) * C + i];
  printf("mcmc %.17g %.17g %lld %lld %lld %lld\n", mean / C, ts[(ITERS - 1) * C], (long long)acc_n, (long long)eval_n,
         (long long)nonfinite, (long long)done);
  printf("backend %s version %d devices %d\n", rsf_backend(), rsf_version(), rsf_device_count());
  /* posterior pool collectives with a one-rank communicator (in the HIP library: a real RCCL communicator) */
  uint8_t id[RSF_COMM_ID_BYTES];
  double pool[ITERS * C], sums[2] = {mean, 1.0};
  CHECK(rsf_comm_unique_id(id));
  CHECK(rsf_comm_init(ctx, 1, 0, id));
  CHECK(rsf_pool_allgather(ctx, tq, ITERS * C, pool));
  CHECK(rsf_pool_allreduce_sum(ctx, sums, 2));
  int same = sums[0] == mean && sums[1] == 1.0;
  for (int i = 0; i < ITERS * C; ++i) same = same && pool[i] == tq[i];
  CHECK(rsf_comm_destroy(ctx));
  printf("pool %s\n", same ? "ok" : "MISMATCH");
  /* TWO ctxs driven by this one thread (rsf_comm_init_all + the grouped collectives): each samples its half of the
   * chains under its chain_offset, and the pooled rows must equal the one-ctx run above.  Two ranks on one device need
   * the test stub (RSF_RCCL_LIB = tests/c/fake_rccl.c); on a multi-GPU node give each ctx its own cfg.device instead. */
  if (getenv("RSF_RCCL_LIB")) {
    enum { H = C / 2 };
    static double tqh[2][ITERS * H], poolh[2][ITERS * C];
    rsf_ctx *cx[2];
    for (int r = 0; r < 2; ++r) {
      CHECK(rsf_create(&cfg, &cx[r]));
      CHECK(rsf_set_model(cx[r], &m));
      rsf_mcmc_config mh = mc;
      mh.n_chains = H;
      mh.chain_offset = r * H;
      CHECK(rsf_mcmc_init(cx[r], &mh, q0 + r * H, data));
      CHECK(rsf_mcmc_run(cx[r], ITERS, tqh[r], NULL, NULL));
    }
    const double *snd[2] = {tqh[0], tqh[1]};
    double *rcv[2] = {poolh[0], poolh[1]};
    double red[2][2] = {{1.0, 2.0}, {10.0, 20.0}};
    double *rb[2] = {red[0], red[1]};
    CHECK(rsf_comm_init_all(cx, 2));
    CHECK(rsf_pool_allgather_all(cx, 2, snd, ITERS * H, rcv));
    CHECK(rsf_pool_allreduce_sum_all(cx, 2, rb, 2));
    int same2 = 1;
    for (int r = 0; r < 2; ++r) {
      same2 = same2 && red[r][0] == 11.0 && red[r][1] == 22.0;
      for (int g = 0; g < 2; ++g)
        for (int it = 0; it < ITERS; ++it)
          for (int i = 0; i < H; ++i) same2 = same2 && rcv[r][g * (ITERS * H) + it * H + i] == tq[it * C + g * H + i];
    }
    for (int r = 0; r < 2; ++r) { CHECK(rsf_comm_destroy(cx[r])); CHECK(rsf_destroy(cx[r])); }
    printf("pool2 %s\n", same2 ? "ok" : "MISMATCH");
  } else {
    printf("pool2 skipped\n");
  }
  CHECK(rsf_destroy(ctx));
  free(acc); free(data);
  return 0;
}
