// tile_rect.h - the rectangle of 16 x 16 tiles an instance is listed in: the reference's square around the 3-sigma radius (gsplat
// isect_tiles through gslam/rasterization.py:259-272) and its tightened form.  Shared by the binning (isect_bin.hip) and the
// projection (project.hip: gsx_project_fwd_rects packs the rectangle for the binning to read instead of means2d + radius).
#pragma once
#include "gsx_common.h"

namespace gsx_rect {

__device__ __forceinline__ uint32_t sat_u32(float f) {
    if (!(f > 0.0f)) return 0u;
    if (f >= 4294967296.0f) return 0xFFFFFFFFu;
    return (uint32_t)f;
}

struct Rect {
    int x0, y0, x1, y1;
};


__device__ __forceinline__ Rect tile_rect(float mx, float my, int32_t radius, int tile_w, int tile_h) {
    const float ts = (float)GSX_TILE;
    const float tr = (float)radius / ts, tx = mx / ts, ty = my / ts;
    Rect r;
    r.x0 = (int)min(sat_u32(floorf(tx - tr)), (uint32_t)tile_w);
    r.y0 = (int)min(sat_u32(floorf(ty - tr)), (uint32_t)tile_h);
    r.x1 = (int)min(sat_u32(ceilf(tx + tr)), (uint32_t)tile_w);
    r.y1 = (int)min(sat_u32(ceilf(ty + tr)), (uint32_t)tile_h);
    return r;
}

// ---- tight rectangle (GSX_PROJ_TILE_EXACT; the fused front of pose-only closures) --------------------------------------------------
// The reference lists an instance in every tile of the square around its 3-sigma radius (gsplat isect_tiles through
// gslam/rasterization.py:259-272) and its rasteriser then skips the instance at every pixel whose alpha = opacity exp(-sigma) stays
// below 1/255.  sigma(d) = 0.5 (a dx^2 + c dy^2) + b dx dy <= L = ln(255 opacity) is an ellipse whose axis-aligned bounding box has
// the half extents sqrt(2 L c / det), sqrt(2 L a / det): a tile none of whose pixel CENTRES lies in that box changes nothing in any
// output of the render or its backward.  The square shrinks to its intersection with the box (never grows): on the headline's map
// 28 % of the (instance, tile) pairs go - the minor-axis side of anisotropic splats, the rim of translucent ones, and whole
// instances whose opacity is below 1/255.  Margins (1 % + 0.02 on L, 0.1 % + 0.01 px on the extents) cover the rounding of the
// rasteriser's own evaluation.  (A per-tile test on top - is the quadratic's minimum over the tile below L? - drops 34 %, but costs
// the projection more than the rasteriser gains: tools/experiments/r05_exact_tile_masks.patch)
__device__ __forceinline__ Rect tighten_rect(Rect r, float mx, float my, float a, float b, float c, float opac) {
    const float l = __logf(255.0f * opac);
    const float two_l = 2.0f * (l + 0.01f * fabsf(l) + 0.02f);
    const float det = a * c - b * b;
    if (!(det > 0.0f) || !(two_l == two_l)) return r;          // degenerate conic / NaN opacity: the reference's square
    if (two_l <= 0.0f) { r.x1 = r.x0; r.y1 = r.y0; return r; } // never reaches 1/255 anywhere
    const float k = two_l / det;
    const float ex = sqrtf(k * c) * 1.001f + 0.01f, ey = sqrtf(k * a) * 1.001f + 0.01f;
    if (!(ex == ex) || !(ey == ey)) return r;
    const float ts = (float)GSX_TILE, inv = 1.0f / (float)GSX_TILE;
    // tile t holds the centres t * ts + 0.5 .. t * ts + ts - 0.5
    const float fx0 = ceilf((mx - ex - (ts - 0.5f)) * inv), fx1 = floorf((mx + ex - 0.5f) * inv) + 1.0f;
    const float fy0 = ceilf((my - ey - (ts - 0.5f)) * inv), fy1 = floorf((my + ey - 0.5f) * inv) + 1.0f;
    r.x0 = max(r.x0, (int)fmaxf(fx0, -1.0e6f)); r.x1 = min(r.x1, (int)fminf(fx1, 1.0e6f));
    r.y0 = max(r.y0, (int)fmaxf(fy0, -1.0e6f)); r.y1 = min(r.y1, (int)fminf(fy1, 1.0e6f));
    if (r.x1 < r.x0) r.x1 = r.x0;
    if (r.y1 < r.y0) r.y1 = r.y0;
    return r;
}

// one word per (camera, Gaussian) for the binning: x0 | x1 << 8 | y0 << 16 | y1 << 24 (tile grids below 256 x 256; 0 = lists nowhere)
__device__ __forceinline__ uint32_t pack_rect(const Rect &r) {
    if (!(r.x1 > r.x0 && r.y1 > r.y0)) return 0u;
    return (uint32_t)r.x0 | ((uint32_t)r.x1 << 8) | ((uint32_t)r.y0 << 16) | ((uint32_t)r.y1 << 24);
}
__device__ __forceinline__ Rect unpack_rect(uint32_t w) {
    Rect r;
    r.x0 = (int)(w & 255u); r.x1 = (int)((w >> 8) & 255u); r.y0 = (int)((w >> 16) & 255u); r.y1 = (int)(w >> 24);
    return r;
}

}  // namespace gsx_rect
