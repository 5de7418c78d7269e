/* The headline measurement from a host WITHOUT Python: the waveform and the spike-timing cVAE (batch 512), each exported with resident
 * tables, stepped side by side on two picked HIP streams through nothing but include/hippie_hip.h — one graph replay per model-step,
 * loader included (hp_model_train_step_staged).
 *
 *   python -m hippie_amd.export --kind unimodal --z-dim 10 --output-size 50  --batch 512 --lr 1e-3 --resident-units 15631 --seed 42 -o wave.hpm
 *   python -m hippie_amd.export --kind unimodal --z-dim 10 --output-size 100 --batch 512 --lr 1e-3 --clip 1.0 --resident-units 15631 --seed 43 -o time.hpm
 *   gcc -std=c99 -O2 -D_POSIX_C_SOURCE=199309L -I include tools/c_host/pair_bench.c -o pair_bench -L hippie_amd -lhippie_hip -Wl,-rpath,$PWD/hippie_amd -Wl,-rpath,/opt/rocm/lib
 *   ./pair_bench wave.hpm time.hpm 300 30
 * Synthetic tables (uniform noise in [-1, 1], labels 1..4): throughput does not depend on the values.  Prints one line. */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <time.h>

#include "hippie_hip.h"

#define CHECK(call)                                                          \
  do {                                                                       \
    if ((call) != 0) {                                                       \
      fprintf(stderr, "%s failed: %s\n", #call, hp_last_error());            \
      return 1;                                                              \
    }                                                                        \
  } while (0)

static uint64_t rng = 88172645463325252ull;
static uint64_t next(void) { rng ^= rng << 13; rng ^= rng >> 7; rng ^= rng << 17; return rng; }

static double now(void) {
  struct timespec t;
  clock_gettime(CLOCK_MONOTONIC, &t);
  return (double)t.tv_sec + 1e-9 * (double)t.tv_nsec;
}

static int fill_tables(HpModel* m, int seed) {
  int32_t cfg[16];
  CHECK(hp_model_config(m, cfg));
  const int64_t N = cfg[10], L = cfg[2];
  if (N <= 0) { fprintf(stderr, "the model was exported without --resident-units\n"); return 1; }
  float* x = (float*)malloc((size_t)(N * L) * 4);
  int64_t* lab = (int64_t*)malloc((size_t)N * 8);
  int64_t* perm = (int64_t*)malloc((size_t)N * 8);
  for (int64_t i = 0; i < N * L; ++i) x[i] = (float)((double)(next() >> 11) / 4503599627370496.0 - 1.0);
  for (int64_t i = 0; i < N; ++i) { lab[i] = 1 + (int64_t)(next() % 4); perm[i] = i; }
  for (int64_t i = N - 1; i > 0; --i) { const int64_t j = (int64_t)(next() % (uint64_t)(i + 1)), t = perm[i]; perm[i] = perm[j]; perm[j] = t; }
  const int64_t s = seed, zero = 0;
  CHECK(hp_model_write(m, "data_x", x, N * L * 4, 0, NULL));
  CHECK(hp_model_write(m, "data_labels", lab, N * 8, 0, NULL));
  CHECK(hp_model_write(m, "perm", perm, N * 8, 0, NULL));
  CHECK(hp_model_write(m, "seed", &s, 8, 0, NULL));
  CHECK(hp_model_write(m, "cursor", &zero, 8, 0, NULL));
  CHECK(hp_model_synchronize(m, NULL));
  free(x); free(lab); free(perm);
  return 0;
}

int main(int argc, char** argv) {
  if (argc < 5) { fprintf(stderr, "usage: %s wave.hpm time.hpm steps warmup\n", argv[0]); return 2; }
  const int steps = atoi(argv[3]), warmup = atoi(argv[4]);
  HpModel* m[2];
  int32_t cfg[16];
  for (int j = 0; j < 2; ++j) {
    CHECK(hp_model_load(argv[1 + j], 0, &m[j]));
    if (fill_tables(m[j], 1234 + j)) return 1;
  }
  CHECK(hp_model_config(m[0], cfg));
  const int B = cfg[7];
  void* s[2];
  float rep[3];
  CHECK(hp_pick_concurrent_streams(m[0], m[1], 6, 0.f, &s[0], &s[1], rep));
  for (int k = 0; k < warmup; ++k)
    for (int j = 0; j < 2; ++j) CHECK(hp_model_train_step_staged(m[j], 1, s[j]));
  for (int j = 0; j < 2; ++j) CHECK(hp_model_synchronize(m[j], s[j]));
  /* the host stays RUN_AHEAD steps ahead of the slower stream (events), as bench.py does: queueing everything at once is ~1 % slower */
  enum { RUN_AHEAD = 2 };
  void* ev[2][RUN_AHEAD];
  for (int j = 0; j < 2; ++j)
    for (int r = 0; r < RUN_AHEAD; ++r) CHECK(hp_event_create(&ev[j][r]));
  const double t0 = now();
  for (int k = 0; k < steps; ++k) {
    if (k >= RUN_AHEAD)
      for (int j = 0; j < 2; ++j) CHECK(hp_event_synchronize(ev[j][k % RUN_AHEAD]));
    for (int j = 0; j < 2; ++j) {
      CHECK(hp_model_train_step_staged(m[j], 1, s[j]));
      CHECK(hp_event_record(ev[j][k % RUN_AHEAD], s[j]));
    }
  }
  for (int j = 0; j < 2; ++j) CHECK(hp_model_synchronize(m[j], s[j]));
  const double dt = now() - t0;
  for (int j = 0; j < 2; ++j)
    for (int r = 0; r < RUN_AHEAD; ++r) CHECK(hp_event_destroy(ev[j][r]));
  float sc[2][4];
  for (int j = 0; j < 2; ++j) CHECK(hp_model_read(m[j], "scalars", sc[j], 16, 0, s[j]));
  printf("{\"host\": \"C99 over include/hippie_hip.h\", \"samples_per_s\": %.1f, \"ms_per_step\": %.4f, \"steps\": %d, \"warmup\": %d, \"batch\": %d, "
         "\"stream_pair_us\": %.1f, \"serial_us\": %.1f, \"pairs_tried\": %d, \"loss\": [%.6g, %.6g]}\n",
         (double)B * steps / dt, dt / steps * 1e3, steps, warmup, B, rep[0], rep[1], (int)rep[2], sc[0][0], sc[1][0]);
  for (int j = 0; j < 2; ++j) { CHECK(hp_stream_destroy(s[j])); CHECK(hp_model_destroy(m[j])); }
  return 0;
}
