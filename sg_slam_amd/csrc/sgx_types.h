// sgx_types.h — small POD types shared by the matcher / pose-optimisation kernels
#pragma once
struct SgxCam { float fx, fy, cx, cy, bf, minX, maxX, minY, maxY; };   // Frame::fx.. mbf, mnMinX.. (src/sg-slam/include/Frame.h)
struct SgxScales { float s[12]; };                                      // per-level table (scale factors or inverse sigma^2)
