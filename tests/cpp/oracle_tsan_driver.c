/* ThreadSanitizer driver for the oracle's threaded paths (tests/test_oracle_sanitizers.py): the row-threaded RNG state
 * creation and trace launch (oracle.c: run_rows), the lazily built jump matrices (pthread_once) reached from several
 * threads at once, and orc_tile_probe called concurrently on one scene and camera, as tests/classification_check.py does.
 * Prints a checksum that the test compares with the production build of the oracle.  TEST INFRASTRUCTURE. */
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../oracle/oracle.h"

enum { W = 48, H = 24, NT = 40 };
static float tris[NT * 12];
static orc_scene scene;
static orc_camera cam;

static void* init_state(void* arg) {          /* first use of the jump table from four threads at once */
  uint32_t s[6];
  orc_rng_init(7, (uint64_t)(size_t)arg * 1000003ull, s);
  return (void*)(size_t)s[1];
}

typedef struct { uint32_t x0, hits; } probe_job;
static void* probe(void* arg) {
  probe_job* j = (probe_job*)arg;
  uint32_t pix[64 * 2];
  for (uint32_t i = 0; i < 64; ++i) { pix[2 * i] = j->x0 + (i & 7u); pix[2 * i + 1] = 8 + (i >> 3); }
  const float lens[10] = {0, 0, 1, 0, 0, 1, -1, 0, 0, -1};
  orc_probe_tri out[NT];
  uint32_t nohit = 0;
  orc_probe_init(out, NT);
  orc_tile_probe(&scene, &cam, W, H, pix, 64, lens, 5, 1, NULL, NULL, out, &nohit);
  for (int i = 0; i < NT; ++i) j->hits += out[i].hits;
  return NULL;
}

int main(void) {
  unsigned long long z = 88172645463325252ull;
  for (int i = 0; i < NT * 12; ++i) {
    z ^= z << 13; z ^= z >> 7; z ^= z << 17;
    const float u = (float)(z >> 40) / 16777216.0f;
    const int c = i % 12, k = c % 4;
    tris[i] = k == 3 ? 0.0f : (k == 2 ? -3.0f - 2.0f * u : -1.5f + 3.0f * u);
  }
  memset(&scene, 0, sizeof scene);
  scene.tris = tris; scene.n_tris = NT;
  const float ang[2] = {0.1f, -0.2f};
  orc_camera_init(&cam, ang, 70.0f, 3.0f, 0.05f);
  pthread_t th[4];
  for (size_t i = 0; i < 4; ++i) pthread_create(&th[i], NULL, init_state, (void*)i);
  for (int i = 0; i < 4; ++i) pthread_join(th[i], NULL);

  orc_frame f;
  f.W = W; f.H = H; f.row0 = 0; f.rows = H;
  f.render = calloc((size_t)W * H * 4, sizeof(float));
  f.counts = calloc((size_t)W * H, sizeof(uint32_t));
  f.rng = calloc((size_t)W * H * 6, sizeof(uint32_t));
  f.image = calloc((size_t)W * H, sizeof(uint32_t));
  orc_frame_rng_init(&f, 3, 4);
  orc_frame_clear(&f);
  orc_trace_launch(&scene, &cam, &f, 3, 1, 4);
  orc_trace_launch(&scene, &cam, &f, 2, 0, 3);
  orc_convert(&f);
  unsigned long long sum = 0;
  for (size_t i = 0; i < (size_t)W * H; ++i) sum += f.image[i];

  probe_job jobs[4];
  for (int i = 0; i < 4; ++i) { jobs[i].x0 = 8u * (uint32_t)i; jobs[i].hits = 0; pthread_create(&th[i], NULL, probe, &jobs[i]); }
  unsigned long long hits = 0;
  for (int i = 0; i < 4; ++i) { pthread_join(th[i], NULL); hits += jobs[i].hits; }
  printf("TSAN_DRIVER_OK %llu %llu\n", sum, hits);
  free(f.render); free(f.counts); free(f.rng); free(f.image);
  return 0;
}
