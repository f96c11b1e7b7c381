// Helpers shared by the MSDA kernels (msda.hip, msda_fused.hip): bilinear tap, level geometry read
// with the window guard, XCD-chunked block order.
#pragma once
#include "common.h"

namespace vah {
namespace msda {

template <typename T>
struct Tap {
    int row[4];     // token index inside the level (clamped to 0 when the corner is invalid)
    bool ok[4];     // corner inside the map and sample inside the gate
    T cw[4];        // hh*hw, hh*lw, lh*hw, lh*lw
    T lh, lw, hh, hw;
};

template <typename T>
__device__ __forceinline__ Tap<T> make_tap(T lx, T ly, int H, int W) {
    Tap<T> t;
    const T h_im = ly * (T)H - (T)0.5;
    const T w_im = lx * (T)W - (T)0.5;
    // strict gate of the reference (cuh:288); NaN locations fail every comparison.
    const bool inside = h_im > (T)-1 && w_im > (T)-1 && h_im < (T)H && w_im < (T)W;
    const T hs = inside ? h_im : (T)0;
    const T ws = inside ? w_im : (T)0;
    const T hf = floor(hs), wf = floor(ws);
    const int h_low = (int)hf, w_low = (int)wf;
    const int h_high = h_low + 1, w_high = w_low + 1;
    t.lh = hs - hf;
    t.lw = ws - wf;
    t.hh = (T)1 - t.lh;
    t.hw = (T)1 - t.lw;
    t.cw[0] = t.hh * t.hw;
    t.cw[1] = t.hh * t.lw;
    t.cw[2] = t.lh * t.hw;
    t.cw[3] = t.lh * t.lw;
    const bool hl = h_low >= 0, hh_ = h_high <= H - 1, wl = w_low >= 0, wh = w_high <= W - 1;
    t.ok[0] = inside && hl && wl;
    t.ok[1] = inside && hl && wh;
    t.ok[2] = inside && hh_ && wl;
    t.ok[3] = inside && hh_ && wh;
    t.row[0] = t.ok[0] ? h_low * W + w_low : 0;
    t.row[1] = t.ok[1] ? h_low * W + w_high : 0;
    t.row[2] = t.ok[2] ? h_high * W + w_low : 0;
    t.row[3] = t.ok[3] ? h_high * W + w_high : 0;
    return t;
}

struct Level {
    int H, W;
    int64_t start;
    bool valid;
};

// Scalar (SGPR) read of one level's geometry, with the window guard described in the ABI.
__device__ __forceinline__ Level read_level(const int64_t *__restrict__ shapes,
                                            const int64_t *__restrict__ lsi, int l, int64_t S) {
    Level lv;
    const int64_t H = shapes[2 * l], W = shapes[2 * l + 1], st = lsi[l];
    lv.valid = H >= 1 && W >= 1 && st >= 0 && H <= S && W <= S && st + H * W <= S;
    lv.H = (int)H;
    lv.W = (int)W;
    lv.start = st;
    return lv;
}

// Each XCD (observed: blockIdx % 8) gets one contiguous chunk of the logical blocks.
// Placement only affects speed, never results.
__device__ __forceinline__ int64_t xcd_chunked_block(int64_t nblocks) {
    const int64_t b = blockIdx.x;
    const int64_t chunk = (nblocks + 7) / 8;
    return (b % 8) * chunk + b / 8;
}

constexpr int kBlock = 256;

}  // namespace msda
}  // namespace vah
