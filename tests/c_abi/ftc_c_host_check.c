/*
 * Host-only walk over the C ABI for a sanitizer build (ASan / UBSan run of the shim): everything include/ftc.h
 * does on the HOST -- checkpoint folding and packing (ftc_create, all four precisions), plan building and validation for several
 * shapes, op introspection, kernel labels, the bounded decoder-plan cache (more row counts than it holds: eviction), error paths
 * (bad arguments, missing tensors, invalid ops) -- without touching a device.  tests/test_c_abi.py builds the library's host
 * translation units (model.hip, ftc_api.hip) and this file with -fsanitize=address,undefined and runs it.
 *
 *   ftc_c_host_check <weights.bin>      (file format of ftc_c_smoke.c)
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "ftc.h"

#define CHECK(c) do { if (!(c)) { fprintf(stderr, "check failed at %s:%d: %s (last error: %s)\n", __FILE__, __LINE__, #c, ftc_last_error()); return 4; } } while (0)

static int read_exact(FILE* f, void* dst, size_t n) { return fread(dst, 1, n, f) == n ? 0 : -1; }

int main(int argc, char** argv) {
    if (argc < 2) { fprintf(stderr, "usage: %s weights.bin\n", argv[0]); return 1; }
    CHECK(ftc_abi_version() == FTC_ABI_VERSION);
    FILE* f = fopen(argv[1], "rb");
    if (!f) { perror(argv[1]); return 1; }
    char magic[4];
    uint32_t n = 0;
    if (read_exact(f, magic, 4) || memcmp(magic, "FTCW", 4) || read_exact(f, &n, 4)) { fprintf(stderr, "bad weights file\n"); return 1; }
    ftc_tensor* tensors = (ftc_tensor*)calloc(n, sizeof(ftc_tensor));
    for (uint32_t i = 0; i < n; ++i) {
        uint32_t len = 0, ndim = 0;
        uint64_t nbytes = 0;
        if (read_exact(f, &len, 4)) return 1;
        char* name = (char*)malloc(len + 1);
        if (read_exact(f, name, len) || read_exact(f, &ndim, 4) || read_exact(f, tensors[i].shape, 32) || read_exact(f, &nbytes, 8)) return 1;
        name[len] = 0;
        void* data = malloc(nbytes ? nbytes : 1);
        if (read_exact(f, data, nbytes)) return 1;
        tensors[i].name = name; tensors[i].data = data; tensors[i].dtype = FTC_F32; tensors[i].ndim = (int32_t)ndim;
    }
    fclose(f);

    /* error paths first */
    ftc_model* bad = NULL;
    CHECK(ftc_create(tensors, (int)n, "xl", 17, &bad) == FTC_ERR_INVALID);
    CHECK(ftc_create(tensors, (int)n / 2, "xl", FTC_F32, &bad) == FTC_ERR_INVALID && strlen(ftc_last_error()) > 0);     /* missing tensors are named */
    CHECK(ftc_create(NULL, 0, "xl", FTC_F32, &bad) != FTC_OK);
    ftc_op junk;
    memset(&junk, 0, sizeof junk);
    junk.kind = 999; junk.B = junk.H = junk.W = 1;
    ftc_plan* jp = NULL;
    CHECK(ftc_plan_create(&junk, 1, 1024, 1024, &jp) == FTC_ERR_INVALID);
    junk.kind = FTC_OP_CONV;                                       /* a convolution with nothing filled in */
    CHECK(ftc_plan_create(&junk, 1, 1024, 1024, &jp) == FTC_ERR_INVALID);

    const int precisions[4] = {FTC_F32, FTC_PRECISION_F16X3, FTC_F16, FTC_BF16};
    long checksum = 0;
    for (int pi = 0; pi < 4; ++pi) {
        ftc_model* m = NULL;
        CHECK(ftc_create(tensors, (int)n, "xl", precisions[pi], &m) == FTC_OK);
        CHECK(ftc_weights_bytes(m) > 400000000 && ftc_weights_host(m) != NULL);
        CHECK(ftc_weights_offset(m, "heads.L0.w") >= 0 && ftc_weights_offset(m, "no.such.tensor") == -1);
        const int shapes[3][3] = {{1, 128, 128}, {2, 256, 192}, {8, 768, 768}};
        for (int si = 0; si < 3; ++si) {
            const int B = shapes[si][0], H = shapes[si][1], W = shapes[si][2];
            CHECK(ftc_workspace_bytes(m, B, H, W) > 0);
            const ftc_plan* plan = NULL;
            ftc_plan_info info;
            CHECK(ftc_model_plan(m, B, H, W, 0, &plan, &info) == FTC_OK && info.n_ops > 250 && info.map_h == H / 4);
            for (int i = 0; i < info.n_ops; ++i) {
                ftc_op op;
                ftc_op_info oi;
                char label[160];
                CHECK(ftc_plan_op(plan, i, &op) == FTC_OK && ftc_model_op_info(m, B, H, W, 0, i, &oi) == FTC_OK);
                CHECK(ftc_op_kernel_label(&op, label, sizeof label) == FTC_OK);
                checksum += op.kind + (long)strlen(label) + (long)(oi.flops / 1e6);
            }
            CHECK(ftc_plan_op(plan, info.n_ops, &junk) != FTC_OK);             /* out of range */
        }
        CHECK(ftc_workspace_bytes(m, 1, 100, 100) == -1);                       /* not a multiple of 32 */
        for (int rows = 1; rows <= 40; ++rows)                                  /* 40 row counts through a cache of 16: eviction */
            CHECK(ftc_decoder_workspace_bytes(m, rows * 37) > 0);
        CHECK(ftc_decoder_workspace_bytes(m, 37) > 0 && ftc_decoder_workspace_bytes(m, 0) == -1);
        ftc_destroy(m);
    }
    ftc_destroy(NULL);
    CHECK(ftc_wgrad_splits(8, 192, 192, 192, 256, 3) >= 1 && ftc_losses_scratch_bytes() > 0 && ftc_decode_scratch_bytes(8, 192, 192) > 0);
    for (uint32_t i = 0; i < n; ++i) { free((void*)tensors[i].name); free((void*)tensors[i].data); }
    free(tensors);
    printf("OK host-side ABI walk, checksum %ld\n", checksum);
    return 0;
}
