/* A plain C99 client of include/wifirx.h: what a compiled host (e.g. a GNU Radio C++ block) would do.
 * Usage: abi_client <iq.c64> <slot_len> <n_slots> <max_sym>
 * Reads interleaved float32 I/Q (GNU Radio file_sink format), runs the batch chain with host buffers and
 * prints one line per slot: flags trigger frame_start encoding psdu_len n_sym_out sum(idx).
 * Exit code 3 when no gfx950 device is usable (the library has no CPU fallback). */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "wifirx.h"

int main(int argc, char** argv)
{
    if (argc < 5) { fprintf(stderr, "usage: %s iq.c64 slot_len n_slots max_sym\n", argv[0]); return 2; }
    const uint32_t slot_len = (uint32_t)atoi(argv[2]), n_slots = (uint32_t)atoi(argv[3]), max_sym = (uint32_t)atoi(argv[4]);
    wifirx_config cfg;
    memset(&cfg, 0, sizeof cfg);
    cfg.abi_version = WIFIRX_ABI_VERSION;
    cfg.bandwidth = 20e6;
    cfg.frequency = 5.89e9;
    cfg.sensitivity = 0.56f;
    cfg.min_plateau = 2;
    cfg.chan_est = WIFIRX_EQ_LS;
    cfg.max_sym = max_sym;
    wifirx_handle* h = NULL;
    int rc = wifirx_create(&cfg, &h);
    if (rc == WIFIRX_ENODEV) { printf("ENODEV %s\n", wifirx_last_error(NULL)); return 3; }
    if (rc) { fprintf(stderr, "create: %d %s\n", rc, wifirx_last_error(NULL)); return 1; }
    size_t n = (size_t)slot_len * n_slots;
    float* iq = (float*)malloc(n * 2 * sizeof(float));
    FILE* f = fopen(argv[1], "rb");
    if (!f || fread(iq, 2 * sizeof(float), n, f) != n) { fprintf(stderr, "short read\n"); return 1; }
    fclose(f);
    wifirx_frame* fr = (wifirx_frame*)calloc(n_slots, sizeof *fr);
    uint8_t* idx = (uint8_t*)calloc((size_t)n_slots * max_sym * 48, 1);
    wifirx_out out;
    memset(&out, 0, sizeof out);
    out.frames = fr;
    out.idx = idx;
    rc = wifirx_demod_batch(h, iq, 0, slot_len, n_slots, &out);
    if (rc) { fprintf(stderr, "demod: %d %s\n", rc, wifirx_last_error(h)); return 1; }
    for (uint32_t i = 0; i < n_slots; i++) {
        unsigned long sum = 0;
        for (uint32_t k = 0; k < (uint32_t)fr[i].n_sym_out * 48; k++) sum += idx[(size_t)i * max_sym * 48 + k];
        printf("%u %d %d %u %u %u %lu\n", fr[i].flags, fr[i].trigger, fr[i].frame_start, fr[i].encoding,
               fr[i].psdu_len, fr[i].n_sym_out, sum);
    }
    wifirx_stats st;
    wifirx_get_stats(h, &st);
    printf("stats %llu %llu\n", (unsigned long long)st.samples_in, (unsigned long long)st.frames_complete);
    wifirx_destroy(h);
    free(iq); free(fr); free(idx);
    return 0;
}
