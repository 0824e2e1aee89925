/* Plain-C caller of the session entry point: what a non-Python host (the Rust crate of
 * INTEGRATION.md, through the same extern "C" declarations) does with libraiko_hip.so.
 *
 *   gcc -O2 -I include examples/session_demo.c -o session_demo -L raiko_amd -lraiko_hip \
 *       -Wl,-rpath,$PWD/raiko_amd
 *   ./session_demo [segments] [po2] [gpu,gpu,...]
 *
 * Builds `segments` synthetic segments (xorshift-filled traces, the tap set of
 * raiko_amd/segment.py:synthetic_tapset for 4/4/12 columns), proves them with rk_prove_session
 * (three in flight per GPU, uploads staged ahead, every seal verified inside the call; with a GPU
 * list all of them take segments from one work queue -- no collective, the seals land in this
 * process's buffers), then checks that a tampered seal is rejected by rk_verify_segment.
 * Exit code 0 on success. */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "raiko_hip.h"

#define P 2013265921u

static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static uint32_t next_elem(void) {
    rng_state ^= rng_state << 13;
    rng_state ^= rng_state >> 7;
    rng_state ^= rng_state << 17;
    return (uint32_t)((rng_state >> 16) % P);
}

enum { W_ACCUM = 4, W_CODE = 4, W_DATA = 12, N_REGS = W_ACCUM + W_CODE + W_DATA };

int main(int argc, char** argv) {
    size_t n_seg = argc > 1 ? (size_t)atoi(argv[1]) : 5;
    unsigned po2 = argc > 2 ? (unsigned)atoi(argv[2]) : 12;
    size_t rows = (size_t)1 << po2;

    /* tap set: combos {0} -> id 0, {0,1} -> id 1, {0,1,2} -> id 2 (sorted, as make_tapset does) */
    static uint32_t combo_off[4] = {0, 1, 3, 6}, combo_backs[6] = {0, 0, 1, 0, 1, 2};
    uint32_t reg_group[N_REGS], reg_offset[N_REGS], reg_combo[N_REGS];
    unsigned r = 0;
    for (unsigned c = 0; c < W_ACCUM; c++, r++) { reg_group[r] = 0; reg_offset[r] = c; reg_combo[r] = 1; }
    for (unsigned c = 0; c < W_CODE; c++, r++) { reg_group[r] = 1; reg_offset[r] = c; reg_combo[r] = 0; }
    for (unsigned c = 0; c < W_DATA; c++, r++) {
        reg_group[r] = 2; reg_offset[r] = c;
        reg_combo[r] = c % 16 == 0 ? 2 : c % 4 == 0 ? 1 : 0;
    }

    rk_segment* segs = calloc(n_seg, sizeof *segs);
    uint32_t** seals = calloc(n_seg, sizeof *seals);
    size_t* caps = calloc(n_seg, sizeof *caps);
    size_t* words = calloc(n_seg, sizeof *words);
    uint32_t globals[8];
    for (int i = 0; i < 8; i++) globals[i] = next_elem();
    const unsigned widths[3] = {W_ACCUM, W_CODE, W_DATA};
    for (size_t s = 0; s < n_seg; s++) {
        rk_segment* g = &segs[s];
        g->po2 = po2;
        g->on_device = 0;
        g->taps.group_size[0] = W_ACCUM; g->taps.group_size[1] = W_CODE; g->taps.group_size[2] = W_DATA;
        g->taps.n_regs = N_REGS;
        g->taps.reg_group = reg_group; g->taps.reg_offset = reg_offset; g->taps.reg_combo = reg_combo;
        g->taps.n_combos = 3; g->taps.combo_off = combo_off; g->taps.combo_backs = combo_backs;
        for (int k = 0; k < 3; k++) {
            uint32_t* m = malloc(rows * widths[k] * 4);
            for (size_t i = 0; i < rows * widths[k]; i++) m[i] = next_elem();
            g->group[k] = m;
        }
        uint32_t* chk = malloc(rows * 16 * 4);
        for (size_t i = 0; i < rows * 16; i++) chk[i] = next_elem();
        g->check = chk;
        g->globals = globals; g->n_globals = 8; g->n_accum_mix = 40;
        memcpy(g->proof_system_info, "RISC0_STARK:v1__", 16);
        memcpy(g->circuit_info, "RV32IM:v1_______", 16);
        caps[s] = rk_seal_bound_words(g);
        seals[s] = malloc(caps[s] * 4);
    }

    int devices[64], n_devices = 0;
    if (argc > 3) {
        for (char* tok = strtok(argv[3], ","); tok && n_devices < 64; tok = strtok(NULL, ",")) devices[n_devices++] = atoi(tok);
    }
    rk_session_opts opts;
    memset(&opts, 0, sizeof opts);
    opts.device = 0;
    opts.inflight = 3;
    opts.upload_ahead = 2;
    opts.verify = 1;
    opts.devices = n_devices ? devices : NULL;
    opts.n_devices = n_devices;
    size_t failed = 0;
    int st = rk_prove_session(&opts, segs, n_seg, seals, caps, words, &failed);
    if (st != RK_OK) {
        fprintf(stderr, "rk_prove_session: %s (%s), segment %zu\n", rk_strerror(st),
                rk_session_last_error(n_devices ? devices[0] : 0), failed);
        return 1;
    }
    for (size_t s = 0; s < n_seg; s++) {
        if (rk_verify_segment(&segs[s], seals[s], words[s]) != 0) return 2;
        seals[s][words[s] / 2] ^= 1u;
        if (rk_verify_segment(&segs[s], seals[s], words[s]) == 0) return 3;
    }
    printf("session_demo: %zu segments of 2^%u cycles proven and verified, seal 0 = %zu words\n", n_seg, po2, words[0]);
    rk_session_release();
    return 0;
}
