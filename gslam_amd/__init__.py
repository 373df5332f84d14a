"""gslam_amd — MI355X-native (gfx950) drop-in for the render / loss / warp hot path of abhigyan7/gslam.

Public surface mirrors the reference:
  gslam_amd.rasterization.rasterization / RasterizationOutput   ≙ gslam/rasterization.py:17-360
  gslam_amd.rendering.rasterization                              ≙ gsplat.rendering.rasterization (pipeline.py:106-116)
  gslam_amd.ops.{fully_fused_projection,isect_tiles,isect_offset_encode,rasterize_to_pixels,...}
                                                                 ≙ gsplat.cuda._wrapper (rasterization.py:9-14)
  gslam_amd.ssim.fused_ssim                                      ≙ fused_ssim (backend.py:13,303-307)
  gslam_amd.warp.Warp                                            ≙ gslam/warp.py:7-82
All compute goes through the C-ABI library gslam_amd/libgsx.so (include/gsx.h); there is no CPU fallback.
"""
__version__ = "0.1.0"
