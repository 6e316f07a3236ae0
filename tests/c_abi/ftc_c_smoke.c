/*
 * C-only client of include/ftc.h (no Python, no C++, no torch): what a non-Python host of the reference's detector path does.
 *
 *   ftc_c_smoke <weights.bin> <case.bin> [bf16]
 *
 * weights.bin : "FTCW" u32 n | per tensor: u32 name_len, name, u32 ndim, i64 shape[4], u64 nbytes, fp32 data   (a state_dict)
 * case.bin    : "FTCC" u32 B, H, W | image [B,H,W,3] fp32 0..1 | golden heatmap [B,10,H/4,W/4] fp32 | golden features [B,100,H/4,W/4]
 *               (the golden arrays are the reference's own CenterNetDetector outputs, tests/golden/g1_fwd128.npz)
 *
 * Steps: ftc_create (fold + pack in the library) -> hipMalloc/hipMemcpy of the packed blob -> ftc_workspace_bytes ->
 * ftc_forward on a stream -> compare with the golden: |diff| < 1e-3 on finite entries, -inf (NMS-suppressed) pattern identical.
 * Exit code 0 and a line "OK ..." on success.
 */
#define __HIP_PLATFORM_AMD__ 1
#include <hip/hip_runtime_api.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "ftc.h"

#define CHECK_HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 2; } } while (0)
#define CHECK_FTC(x) do { int r_ = (x); if (r_ != FTC_OK) { fprintf(stderr, "ftc error %d at %s:%d: %s\n", r_, __FILE__, __LINE__, ftc_last_error()); return 3; } } while (0)

static int read_exact(FILE* f, void* dst, size_t n) { return fread(dst, 1, n, f) == n ? 0 : -1; }

int main(int argc, char** argv) {
    if (argc < 3) { fprintf(stderr, "usage: %s weights.bin case.bin [bf16]\n", argv[0]); return 1; }
    const int precision = (argc > 3 && strcmp(argv[3], "bf16") == 0) ? FTC_BF16 : FTC_F32;
    if (ftc_abi_version() != FTC_ABI_VERSION) { fprintf(stderr, "ABI mismatch: header %d, library %d\n", FTC_ABI_VERSION, ftc_abi_version()); return 1; }
    int n_cu = 0;
    char arch[64];
    CHECK_FTC(ftc_device_info(&n_cu, arch, sizeof arch));           /* fails loudly (exit 3) when no gfx950 device is visible */

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
        if (read_exact(f, data, nbytes)) { fprintf(stderr, "truncated tensor %s\n", name); return 1; }
        tensors[i].name = name; tensors[i].data = data; tensors[i].dtype = FTC_F32; tensors[i].ndim = (int32_t)ndim;
    }
    fclose(f);

    f = fopen(argv[2], "rb");
    if (!f) { perror(argv[2]); return 1; }
    uint32_t dims[3];
    if (read_exact(f, magic, 4) || memcmp(magic, "FTCC", 4) || read_exact(f, dims, 12)) { fprintf(stderr, "bad case file\n"); return 1; }
    const int B = (int)dims[0], H = (int)dims[1], W = (int)dims[2], h = H / 4, w = W / 4;
    const size_t n_img = (size_t)B * H * W * 3, n_heat = (size_t)B * 10 * h * w, n_feat = (size_t)B * 100 * h * w;
    float* img = (float*)malloc(n_img * 4);
    float* g_heat = (float*)malloc(n_heat * 4);
    float* g_feat = (float*)malloc(n_feat * 4);
    if (read_exact(f, img, n_img * 4) || read_exact(f, g_heat, n_heat * 4) || read_exact(f, g_feat, n_feat * 4)) { fprintf(stderr, "truncated case file\n"); return 1; }
    fclose(f);

    ftc_model* model = NULL;
    CHECK_FTC(ftc_create(tensors, (int)n, "xl", precision, &model));
    for (uint32_t i = 0; i < n; ++i) { free((void*)tensors[i].data); free((void*)tensors[i].name); }   /* may be freed after ftc_create */
    free(tensors);

    void *d_w = NULL, *d_ws = NULL, *d_img = NULL, *d_heat = NULL, *d_feat = NULL;
    const int64_t wbytes = ftc_weights_bytes(model), wsbytes = ftc_workspace_bytes(model, B, H, W);
    if (wsbytes < 0) { fprintf(stderr, "ftc_workspace_bytes: %s\n", ftc_last_error()); return 3; }
    CHECK_HIP(hipMalloc(&d_w, (size_t)wbytes));
    CHECK_HIP(hipMemcpy(d_w, ftc_weights_host(model), (size_t)wbytes, hipMemcpyHostToDevice));
    CHECK_HIP(hipMalloc(&d_ws, (size_t)wsbytes));
    CHECK_HIP(hipMalloc(&d_img, n_img * 4));
    CHECK_HIP(hipMalloc(&d_heat, n_heat * 4));
    CHECK_HIP(hipMalloc(&d_feat, n_feat * 4));
    CHECK_HIP(hipMemcpy(d_img, img, n_img * 4, hipMemcpyHostToDevice));
    hipStream_t stream;
    CHECK_HIP(hipStreamCreate(&stream));
    CHECK_FTC(ftc_forward(model, d_w, d_img, B, H, W, /*nchw=*/0, /*with_nms=*/1, d_heat, d_feat, d_ws, stream));
    CHECK_HIP(hipStreamSynchronize(stream));
    float* heat = (float*)malloc(n_heat * 4);
    float* feat = (float*)malloc(n_feat * 4);
    CHECK_HIP(hipMemcpy(heat, d_heat, n_heat * 4, hipMemcpyDeviceToHost));
    CHECK_HIP(hipMemcpy(feat, d_feat, n_feat * 4, hipMemcpyDeviceToHost));

    /* library output is NHWC, the golden is NCHW */
    double e_heat = 0.0, e_feat = 0.0;
    long inf_mismatch = 0, n_inf = 0;
    for (int b = 0; b < B; ++b)
        for (int y = 0; y < h; ++y)
            for (int x = 0; x < w; ++x) {
                for (int c = 0; c < 10; ++c) {
                    const float a = heat[(((size_t)b * h + y) * w + x) * 10 + c], r = g_heat[(((size_t)b * 10 + c) * h + y) * w + x];
                    if (isinf(r) || isinf(a)) { n_inf += isinf(r) ? 1 : 0; if (!(isinf(r) && isinf(a) && r < 0 && a < 0)) ++inf_mismatch; continue; }
                    const double d = fabs((double)a - (double)r);
                    if (!(d <= e_heat)) e_heat = d;                      /* also catches NaN */
                }
                for (int c = 0; c < 100; ++c) {
                    const double d = fabs((double)feat[(((size_t)b * h + y) * w + x) * 100 + c] - (double)g_feat[(((size_t)b * 100 + c) * h + y) * w + x]);
                    if (!(d <= e_feat)) e_feat = d;
                }
            }
    const double tol = precision == FTC_F32 ? 1e-3 : 0.5;
    const long allowed_inf_mismatch = precision == FTC_F32 ? 0 : n_inf / 10;
    ftc_destroy(model);
    hipFree(d_w); hipFree(d_ws); hipFree(d_img); hipFree(d_heat); hipFree(d_feat);
    if (!(e_heat < tol) || !(e_feat < tol) || inf_mismatch > allowed_inf_mismatch) {
        printf("FAIL %s heatmap_linf %.3e features_linf %.3e nms_mask_mismatch %ld of %ld suppressed (tol %.1e)\n", arch, e_heat, e_feat, inf_mismatch, n_inf, tol);
        return 4;
    }
    printf("OK %s (%d CUs) %s B=%d %dx%d heatmap_linf %.3e features_linf %.3e nms_mask_mismatch %ld of %ld suppressed\n", arch, n_cu,
           precision == FTC_F32 ? "fp32" : "bf16", B, H, W, e_heat, e_feat, inf_mismatch, n_inf);
    return 0;
}
