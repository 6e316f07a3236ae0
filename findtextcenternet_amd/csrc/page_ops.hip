// Page-level front/back ends of the detector path (SURVEY.md 8f "next" rows 1 and 3):
//  * tile_gather : the tiling front-end of OCR_Processer.call_OCR (/root/reference/process_ocr_base.py:67-76)
//                  -- uint8 page in HBM -> batch of [768,768,3] float tiles scaled by 1/255
//                  (process_ocr_torch.py:44), so a page costs one H2D copy of its uint8 pixels;
//  * paste_maps  : the np.maximum paste of the masked sigmoid maps into the page canvases
//                  (/root/reference/process_ocr_base.py:505-516), on the GPU so that only the
//                  page-sized canvases -- not every tile's maps -- go back to the host.
#include "ftc_common.h"

namespace {

__global__ __launch_bounds__(256) void tile_gather_kernel(const unsigned char* __restrict__ page, int PH, int PW,
                                                          const int* __restrict__ origins, int th, int tw,
                                                          float* __restrict__ out) {
    const int b = blockIdx.z;
    const int oy = origins[2 * b], ox = origins[2 * b + 1];
    const long n = (long)th * tw * 3;
    float* ob = out + (long)b * n;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % 3);
        const int x = (int)((i / 3) % tw);
        const int y = (int)(i / (3L * tw));
        const int py = oy + y, px = ox + x;
        // pixels beyond the page are the reference's white padding (np.pad ... 255, process_ocr_base.py:65)
        const float v = (py < PH && px < PW) ? (float)page[((long)py * PW + px) * 3 + c] : 255.0f;
        ob[i] = v / 255.0f;                                   // IEEE division: same bits as numpy float32 x / 255.
    }
}

__device__ __forceinline__ float ref_sigmoid(float x) { return (tanhf(x * 0.5f) + 1.0f) * 0.5f; }   // util_func.py:15

// canvases: [7][ph][pw] fp32 (key, line, sep, code1, code2, code4, code8), zero-initialised by the caller.
// Values are >= 0, so max on the raw bit pattern as int is max on the float: atomicMax gives the
// order-independent (deterministic) np.maximum merge of overlapping tiles.
__global__ __launch_bounds__(256) void paste_maps_kernel(const float* __restrict__ heat, const ftc_tile* __restrict__ tiles,
                                                         int h, int w, int scale, float* __restrict__ canv, int ph, int pw) {
    const int b = blockIdx.y;
    const ftc_tile tl = tiles[b];
    const int rw = tl.x_max - tl.x_min, rh = tl.y_max - tl.y_min;
    const int x_is = tl.offset_x / scale, y_is = tl.offset_y / scale;
    const float* hb = heat + (long)b * h * w * 10;
    const int chan[7] = {0, 4, 5, 6, 7, 8, 9};
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < rw * rh; i += gridDim.x * blockDim.x) {
        const int y = tl.y_min + i / rw, x = tl.x_min + i % rw;
        const int py = y_is + y, px = x_is + x;
        if (py >= ph || px >= pw) continue;
        const float* px10 = hb + ((long)y * w + x) * 10;
#pragma unroll
        for (int k = 0; k < 7; ++k) {
            const float p = ref_sigmoid(px10[chan[k]]);
            atomicMax(reinterpret_cast<int*>(canv + ((long)k * ph + py) * pw + px), __float_as_int(p));
        }
    }
}

}  // namespace

hipError_t launch_tile_gather(const unsigned char* page, int PH, int PW, const int* origins, int B, int th, int tw, float* out,
                              hipStream_t s) {
    hipLaunchKernelGGL(tile_gather_kernel, dim3(1024, 1, B), dim3(256), 0, s, page, PH, PW, origins, th, tw, out);
    return hipGetLastError();
}

hipError_t launch_paste_maps(const float* heat, const ftc_tile* tiles, int B, int h, int w, int scale, float* canv, int ph, int pw,
                             hipStream_t s) {
    hipLaunchKernelGGL(paste_maps_kernel, dim3(36, B), dim3(256), 0, s, heat, tiles, h, w, scale, canv, ph, pw);
    return hipGetLastError();
}
