/* A host WITHOUT Python: loads a serialised lowered model (.hpm, written by hippie_amd/export.py), feeds it a batch from a
 * raw binary file and takes optimisation steps through the reference's verbs — forward / backward / optimizer step
 * (hippie/model.py:95-116 under Lightning's automatic optimisation) — using nothing but include/hippie_hip.h.
 *
 *   host_step model.hpm inputs.bin n_steps use_graph [dp]
 *
 * dp: data-parallel form at world size 1 — hp_dp_unique_id, hp_model_allreduce_init (RCCL bound at run time) and every step through
 * hp_model_train_step_dp (Lightning's DDP strategy around the same training_step, scripts/train_model_with_multimodal.py:200-207).
 *
 * inputs.bin: x float32[B*L] | src int64[B] | eps float32[B*z]  (multimodal: x | x2 | src | eps).
 * Prints one line per step:  step k loss mse1 mse2 kl   and finally checksums of `enc_train` and of the parameter arena.
 * Plain C99: no HIP headers, no C++ — what a cgo / JNI / ctypes stub would bind.  tests/test_gpu_c_host.py builds and runs it. */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "hippie_hip.h"

#define CHECK(call)                                                          \
  do {                                                                       \
    if ((call) != 0) {                                                       \
      fprintf(stderr, "%s failed: %s\n", #call, hp_last_error());            \
      return 1;                                                              \
    }                                                                        \
  } while (0)

static int feed(HpModel* m, const char* slot, FILE* f) {
  HpTensorInfo t;
  if (hp_model_find(m, slot, &t) != 0) return 1;
  const int64_t nbytes = t.numel * (t.dtype == 0 ? 4 : 8);
  void* host = malloc((size_t)nbytes);
  if (host == NULL || fread(host, 1, (size_t)nbytes, f) != (size_t)nbytes) {
    fprintf(stderr, "inputs file too short for slot %s (%lld bytes)\n", slot, (long long)nbytes);
    free(host);
    return 1;
  }
  const int rc = hp_model_write(m, slot, host, nbytes, 0, NULL);
  if (rc == 0) hp_model_synchronize(m, NULL);       /* the copy reads `host` asynchronously */
  free(host);
  return rc;
}

int main(int argc, char** argv) {
  if (argc < 5) {
    fprintf(stderr, "usage: %s model.hpm inputs.bin n_steps use_graph\n", argv[0]);
    return 2;
  }
  const int n_steps = atoi(argv[3]), use_graph = atoi(argv[4]);
  const int dp = argc > 5 && strcmp(argv[5], "dp") == 0;
  HpModel* m = NULL;
  CHECK(hp_model_load(argv[1], 0, &m));
  int32_t cfg[16];
  CHECK(hp_model_config(m, cfg));
  const int multimodal = cfg[0], z = cfg[1], B = cfg[7];
  printf("model kind %d z %d L %d L2 %d batch %d params %d\n", cfg[0], z, cfg[2], cfg[3], B, hp_model_tensor_count(m, 0));

  FILE* f = fopen(argv[2], "rb");
  if (f == NULL) { fprintf(stderr, "cannot open %s\n", argv[2]); return 1; }
  CHECK(feed(m, "x", f));
  if (multimodal) CHECK(feed(m, "x2", f));
  CHECK(feed(m, "src", f));
  CHECK(feed(m, "eps", f));
  fclose(f);

  if (dp) {
    /* a real launcher hands rank 0's id to the other ranks (file, socket, MPI); here the world is this one process */
    char id[HP_DP_UNIQUE_ID_BYTES];
    CHECK(hp_dp_unique_id(id));
    CHECK(hp_model_allreduce_init(m, id, 0, 1));
    printf("data parallel: rank 0 of 1\n");
  }
  for (int k = 0; k < n_steps; ++k) {
    /* the three verbs separately on even steps, the one-graph step on odd ones: both routes are exercised */
    if (dp) {
      CHECK(hp_model_train_step_dp(m, use_graph, NULL));
    } else if (k % 2 == 0) {
      CHECK(hp_model_forward(m, 1, use_graph, NULL));
      CHECK(hp_model_backward(m, use_graph, NULL));
      CHECK(hp_model_optimizer_step(m, use_graph, NULL));
    } else {
      CHECK(hp_model_train_step(m, use_graph, NULL));
    }
    float sc[4];
    CHECK(hp_model_read(m, "scalars", sc, sizeof sc, 0, NULL));
    printf("step %d %.9g %.9g %.9g %.9g\n", k, sc[0], sc[1], sc[2], sc[3]);
  }

  HpTensorInfo enc;
  CHECK(hp_model_find(m, "enc_train", &enc));
  float* e = (float*)malloc((size_t)enc.numel * 4);
  CHECK(hp_model_read(m, "enc_train", e, enc.numel * 4, 0, NULL));
  double s = 0.0, s2 = 0.0;
  for (int64_t i = 0; i < enc.numel; ++i) { s += e[i]; s2 += (double)e[i] * e[i]; }
  printf("enc_train sum %.9g sumsq %.9g\n", s, s2);
  free(e);

  /* one parameter by its reference state_dict key, straight out of the arena */
  HpTensorInfo w;
  CHECK(hp_model_find(m, multimodal ? "encoder_mod1.conv1.weight" : "encoder.conv1.weight", &w));
  float* wv = (float*)malloc((size_t)w.numel * 4);
  CHECK(hp_model_read(m, w.name, wv, w.numel * 4, 0, NULL));
  double ws = 0.0;
  for (int64_t i = 0; i < w.numel; ++i) ws += wv[i];
  printf("%s sum %.9g numel %lld batches_tracked %lld\n", w.name, ws, (long long)w.numel, (long long)hp_model_batches_tracked(m));
  free(wv);
  CHECK(hp_model_destroy(m));
  return 0;
}
