/* A host WITHOUT Python that holds TWO models — the waveform cVAE and the spike-timing cVAE the reference fits one after the other
 * (scripts/train_model_with_multimodal.py:208,224) — and steps them side by side on two HIP streams chosen by
 * hp_pick_concurrent_streams (which pair of streams really overlaps is measured, include/hippie_hip.h).
 *
 *   host_pair wave.hpm wave_inputs.bin time.hpm time_inputs.bin n_steps
 *
 * inputs: x float32[B*L] | src int64[B] | eps float32[B*z] per model.  Prints "pick ..." then, per step and model,
 *   model j step k loss mse1 mse2 kl
 * Plain C99, only include/hippie_hip.h.  tests/test_model_file.py builds and runs it. */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

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
    fprintf(stderr, "inputs file too short for slot %s\n", slot);
    free(host);
    return 1;
  }
  const int rc = hp_model_write(m, slot, host, nbytes, 0, NULL);
  if (rc == 0) hp_model_synchronize(m, NULL);
  free(host);
  return rc;
}

int main(int argc, char** argv) {
  if (argc < 6) {
    fprintf(stderr, "usage: %s wave.hpm wave_inputs.bin time.hpm time_inputs.bin n_steps\n", argv[0]);
    return 2;
  }
  const int n_steps = atoi(argv[5]);
  HpModel* m[2] = {NULL, NULL};
  for (int j = 0; j < 2; ++j) {
    CHECK(hp_model_load(argv[1 + 2 * j], 0, &m[j]));
    FILE* f = fopen(argv[2 + 2 * j], "rb");
    if (f == NULL) { fprintf(stderr, "cannot open %s\n", argv[2 + 2 * j]); return 1; }
    CHECK(feed(m[j], "x", f));
    CHECK(feed(m[j], "src", f));
    CHECK(feed(m[j], "eps", f));
    fclose(f);
  }
  void* s[2] = {NULL, NULL};
  float rep[3];
  CHECK(hp_pick_concurrent_streams(m[0], m[1], 6, 0.f, &s[0], &s[1], rep));
  printf("pick pair_us %.1f serial_us %.1f tried %d distinct %d\n", rep[0], rep[1], (int)rep[2], s[0] != s[1] && s[0] != NULL && s[1] != NULL);

  /* a model's step is queued on ITS stream; reading its scalars back waits for that stream only, so the other model keeps running */
  float* sc = (float*)malloc((size_t)n_steps * 2 * 4 * sizeof(float));
  for (int k = 0; k < n_steps; ++k)
    for (int j = 0; j < 2; ++j) {
      CHECK(hp_model_train_step(m[j], 1, s[j]));
      CHECK(hp_model_read(m[j], "scalars", sc + (k * 2 + j) * 4, 16, 0, s[j]));
    }
  for (int j = 0; j < 2; ++j) CHECK(hp_model_synchronize(m[j], s[j]));
  for (int k = 0; k < n_steps; ++k)
    for (int j = 0; j < 2; ++j) {
      const float* v = sc + (k * 2 + j) * 4;
      printf("model %d step %d %.9g %.9g %.9g %.9g\n", j, k, v[0], v[1], v[2], v[3]);
    }
  free(sc);
  for (int j = 0; j < 2; ++j) {
    printf("model %d batches_tracked %lld\n", j, (long long)hp_model_batches_tracked(m[j]));
    CHECK(hp_stream_destroy(s[j]));
    CHECK(hp_model_destroy(m[j]));
  }
  void* extra = NULL;
  CHECK(hp_stream_create(&extra));
  CHECK(hp_stream_destroy(extra));
  return 0;
}
