"""The boundary against the reference's REAL headers (container only: /root/reference does not travel).

INTEGRATION.md says HipVoxelRayTracer compiles inside the reference tree against IRayTracer.h / VoxelGrid.h /
RayTracerCommon.h / Voxels.h (none of which needs spdlog).  This compiles it that way -- -DHVRT_USE_REFERENCE_HEADERS
selects the reference's headers instead of this repository's restatement -- with the layout static_asserts of
HipVoxelRayTracer.cpp (Voxel 28, Ray 24, RayHit 52, Camera 44 bytes; IRayTracer abstract, HipVoxelRayTracer a
concrete subclass), and a factory hunk shaped like TraceImages.cpp:464-484.  Reference: IRayTracer.h:17-44,
VoxelGrid.h:8-42."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference/RayTracer"
HOST = os.path.join(ROOT, "raytracer_amd", "host")

pytestmark = pytest.mark.skipif(not os.path.isdir(REF), reason="the reference tree exists only in the authoring container")


def _compile(src, out, extra=()):
    cmd = ["g++", "-std=c++20", "-c", "-Wall", "-DHVRT_USE_REFERENCE_HEADERS", "-I" + REF, "-I" + os.path.join(ROOT, "include"),
           "-I" + HOST, *extra, src, "-o", out]
    p = subprocess.run(cmd, capture_output=True, text=True)
    assert p.returncode == 0, p.stderr[-3000:]


def test_hip_voxel_ray_tracer_compiles_against_the_reference_headers(tmp_path):
    _compile(os.path.join(HOST, "HipVoxelRayTracer.cpp"), str(tmp_path / "tracer.o"))
    # the object really was built from the reference's types: its symbols name cg_datastructures::Camera etc.
    syms = subprocess.run(["nm", "-C", str(tmp_path / "tracer.o")], capture_output=True, text=True).stdout
    assert "HipVoxelRayTracer::render(cg_datastructures::Camera const&, int, int)" in syms
    assert "HipVoxelRayTracer::traceRay(cg_datastructures::Ray const&)" in syms


def test_factory_hunk_of_integration_md_compiles(tmp_path):
    """The three-hunk patch of INTEGRATION.md, as a translation unit against the reference's headers: a new enumerator
    and a factory case returning std::unique_ptr<IRayTracer>; the reference's VoxelGrid has no default constructor
    (VoxelGrid.h:18-35), so the grid arrives by reference exactly as in TraceImages::createRayTracer."""
    src = tmp_path / "factory.cpp"
    src.write_text('''
#include <memory>
#include "IRayTracer.h"
#include "VoxelGrid.h"
#include "HipVoxelRayTracer.hpp"
enum class RayTracingMethod { VOXEL_DDA, ART, OCTREE, HIP_VOXEL };
std::unique_ptr<IRayTracer> createRayTracer(VoxelGrid& voxelGrid, RayTracingMethod method) {
    switch (method) {
        case RayTracingMethod::HIP_VOXEL: return std::make_unique<HipVoxelRayTracer>(voxelGrid);
        default: return nullptr;
    }
}
std::vector<cg_datastructures::Vec3> run(VoxelGrid& g, const cg_datastructures::Camera& cam) {
    auto rt = createRayTracer(g, RayTracingMethod::HIP_VOXEL);
    cg_datastructures::RayHit h = rt->traceRay(cg_datastructures::Ray(cam.position, cam.lookAt));
    (void)h; (void)rt->getMethodName();
    return rt->render(cam, 800, 600);
}
static_assert(!std::is_default_constructible<VoxelGrid>::value, "the reference VoxelGrid has no default constructor");
''')
    _compile(str(src), str(tmp_path / "factory.o"))
