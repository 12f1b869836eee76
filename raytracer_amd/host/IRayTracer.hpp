// IRayTracer.hpp -- the reference's plug-in interface (IRayTracer.h:17-44), restated: three pure
// virtuals.  HipVoxelRayTracer implements it; in the reference tree the reference's own header is used.
#pragma once

#include <string>
#include <vector>

#include "rt_types.hpp"

class IRayTracer {
public:
    virtual ~IRayTracer() = default;
    // W*H pixels, row-major y*W+x, row 0 = bottom of the view, components nominally 0..1 (up to 1.1)
    virtual std::vector<cg_datastructures::Vec3> render(const cg_datastructures::Camera& camera, int width, int height) = 0;
    virtual cg_datastructures::RayHit traceRay(const cg_datastructures::Ray& ray) = 0;
    virtual std::string getMethodName() const = 0;
};
