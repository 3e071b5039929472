/* asan_driver.c -- runs the CPU restatement (aft_oracle.c, TEST INFRASTRUCTURE) under
 * AddressSanitizer + UBSan on one serialized case (CPU sanitizer run; GPU ASan is
 * not available on the pool).  Built by `make -C oracle asan_driver`, driven by
 * tests/test_oracle_asan.py.
 *
 *   asan_driver <case.bin> <out.bin>
 *
 * case.bin : aft_config | int32 batch | int32 adaptive-meta flag | uint64 n_floats | float blob[n_floats]
 *            | aft_weights whose "pointers" are (offset into blob + 1), 0 = NULL (its `layers` slot is 0)
 *            | aft_layer_weights[num_layers], same encoding (ABI 7: the layer table is a host array behind a pointer)
 *            | uint64 offsets(+1) of pilots, snr, ds, dop inside the blob
 * out.bin  : complex64 [batch, S, T] as floats
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../include/adafortitran_amd.h"

typedef struct aft_oracle_dump aft_oracle_dump;
int aft_oracle_forward_f32(const aft_config *c, const aft_weights *w, const float *pilots, const float *snr,
                           const float *ds, const float *dop, float *out, int batch, const aft_oracle_dump *dump);
int aft_oracle_mse_partial_f32(const float *est, const float *ref, double *sum_sq, long long n_complex);

static void need(int ok, const char *what) {
    if (!ok) {
        fprintf(stderr, "asan_driver: %s\n", what);
        exit(2);
    }
}

int main(int argc, char **argv) {
    need(argc == 3, "usage: asan_driver case.bin out.bin");
    FILE *f = fopen(argv[1], "rb");
    need(f != NULL, "cannot open case file");
    aft_config cfg;
    int32_t batch = 0, has_meta = 0;
    uint64_t n = 0;
    need(fread(&cfg, sizeof(cfg), 1, f) == 1 && fread(&batch, 4, 1, f) == 1 && fread(&has_meta, 4, 1, f) == 1 &&
             fread(&n, 8, 1, f) == 1, "short header");
    float *blob = (float *)malloc(n * sizeof(float));   /* exact size: an overrun is an ASan report */
    need(blob != NULL && fread(blob, sizeof(float), n, f) == n, "short blob");
    aft_weights w;
    need(fread(&w, sizeof(w), 1, f) == 1, "short weight table");
    need(cfg.num_layers > 0 && cfg.num_layers < (1 << 20), "bad layer count");
    aft_layer_weights *layers = (aft_layer_weights *)malloc((size_t)cfg.num_layers * sizeof(aft_layer_weights));   /* exact size */
    need(layers != NULL && fread(layers, sizeof(aft_layer_weights), (size_t)cfg.num_layers, f) == (size_t)cfg.num_layers, "short layer table");
    uint64_t io[4];
    need(fread(io, 8, 4, f) == 4, "short io table");
    fclose(f);
    /* offsets (+1) -> pointers */
    w.layers = NULL;
    uintptr_t *slots = (uintptr_t *)&w;
    for (size_t i = 0; i < sizeof(w) / sizeof(uintptr_t); ++i)
        if (slots[i]) {
            need(slots[i] - 1 < n, "weight offset out of range");
            slots[i] = (uintptr_t)(blob + (slots[i] - 1));
        }
    slots = (uintptr_t *)layers;
    for (size_t i = 0; i < (size_t)cfg.num_layers * sizeof(aft_layer_weights) / sizeof(uintptr_t); ++i)
        if (slots[i]) {
            need(slots[i] - 1 < n, "layer weight offset out of range");
            slots[i] = (uintptr_t)(blob + (slots[i] - 1));
        }
    w.layers = layers;
    const float *in[4];
    for (int i = 0; i < 4; ++i) in[i] = io[i] ? blob + (io[i] - 1) : NULL;
    const size_t out_floats = (size_t)batch * cfg.num_scs * cfg.num_symbols * 2;
    float *out = (float *)malloc(out_floats * sizeof(float));
    need(out != NULL, "oom");
    const int rc = aft_oracle_forward_f32(&cfg, &w, in[0], has_meta ? in[1] : NULL, has_meta ? in[2] : NULL,
                                          has_meta ? in[3] : NULL, out, batch, NULL);
    need(rc == 0, "oracle returned an error code");
    double sum = 0.0;   /* also walk the metric restatement: est vs itself = 0 */
    need(aft_oracle_mse_partial_f32(out, out, &sum, (long long)(out_floats / 2)) == 0 && sum == 0.0, "metric");
    f = fopen(argv[2], "wb");
    need(f != NULL && fwrite(out, sizeof(float), out_floats, f) == out_floats, "cannot write output");
    fclose(f);
    free(out);
    free(layers);
    free(blob);
    return 0;
}
