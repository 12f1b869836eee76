// trace_image -- command-line front end of the host mirror (the reference's main() has hard-coded
// Windows paths, RayTracer.cpp:8-44; this takes them as arguments).
//   trace_image render <in.stl|in.voxgrid> <out.png|.ppm> [--method hip|voxel_dda|art|octree] [--multiview]
//                      [--grid N] [--size WxH] [--outdir DIR] [--dump-raw FILE] [--nearest] [--quiet]
//                      [--repeat N] [--phases FILE]
//                      (--repeat: the whole TraceImage call N times, as the reference's main() does for its list of models,
//                       RayTracer.cpp:20-40; --phases: one JSON line of per-phase milliseconds per call)
//   trace_image stl-info <in.stl> <out.bin>        (CPU only: vertex / triangle numbering of the loader)
//   trace_image save-voxgrid <in.stl> <out.voxgrid> [--grid N]
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <string>

#include "Logger.hpp"
#include "TraceImages.hpp"

static int usage() {
    std::fprintf(stderr, "usage: trace_image render|stl-info|save-voxgrid ... (see trace_image_main.cpp)\n");
    return 2;
}

int main(int argc, char** argv) {
    if (argc < 4) return usage();
    const std::string cmd = argv[1], in = argv[2], out = argv[3];
    RayTracingMethod method = RayTracingMethod::HIP_VOXEL;
    bool multiview = false, nearest = false;
    int grid = 50, W = 800, H = 600, repeat = 1;
    std::string outdir, dump, phases;
    for (int i = 4; i < argc; ++i) {
        std::string a = argv[i];
        auto next = [&]() -> std::string { if (i + 1 >= argc) { usage(); std::exit(2); } return argv[++i]; };
        if (a == "--method") {
            std::string m = next();
            if (m == "hip") method = RayTracingMethod::HIP_VOXEL;
            else if (m == "voxel_dda") method = RayTracingMethod::VOXEL_DDA;
            else if (m == "art") method = RayTracingMethod::ART;
            else if (m == "octree") method = RayTracingMethod::OCTREE;
            else return usage();
        } else if (a == "--multiview") multiview = true;
        else if (a == "--grid") grid = std::atoi(next().c_str());
        else if (a == "--size") { if (std::sscanf(next().c_str(), "%dx%d", &W, &H) != 2) return usage(); }
        else if (a == "--outdir") outdir = next();
        else if (a == "--dump-raw") dump = next();
        else if (a == "--nearest") nearest = true;
        else if (a == "--repeat") repeat = std::atoi(next().c_str());
        else if (a == "--phases") phases = next();
        else if (a == "--quiet") Logger::setLevel(hvrt_log::ERROR);
        else return usage();
    }
    try {
        if (cmd == "stl-info") {
            stl_reader::StlMesh<float, unsigned int> mesh(in);
            std::ofstream f(out, std::ios::binary);
            unsigned long long nv = mesh.num_vrts(), nt = mesh.num_tris();
            f.write((const char*)&nv, 8);
            f.write((const char*)&nt, 8);
            for (size_t i = 0; i < nv; ++i) f.write((const char*)mesh.vrt_coords(i), 12);
            for (size_t t = 0; t < nt; ++t) f.write((const char*)mesh.tri_corner_inds(t), 12);
            for (size_t t = 0; t < nt; ++t) f.write((const char*)mesh.tri_normal(t), 12);
            return f ? 0 : 1;
        }
        if (cmd == "save-voxgrid") {
            stl_reader::StlMesh<float, unsigned int> mesh(in);
            HipScene scene(0);
            TraceImages::buildSceneOnGpu(mesh, grid, grid, grid, scene);
            VoxelGrid g = scene.downloadGrid();
            return g.saveToFile(out) ? 0 : 1;
        }
        if (cmd == "render") {
            TraceImages ti;
            ti.gridResolution = grid;
            ti.imageWidth = W;
            ti.imageHeight = H;
            ti.outputDirectory = outdir;
            ti.nearestHit = nearest;
            ti.keepHitIds = !dump.empty();      // the raw dump holds the float frame and the hit-id image
            ti.keepPixels = !dump.empty();
            std::ofstream pf;
            if (!phases.empty()) pf.open(phases);
            for (int r = 0; r < (repeat < 1 ? 1 : repeat); ++r) {
                if (multiview) ti.TraceImageMultiView(in, out, method);
                else ti.TraceImage(in, out, method);
                if (!ti.lastOk) break;
                const TraceImages::Phases& p = ti.lastPhases;
                if (pf) pf << "{\"call\": " << r << ", \"context_ms\": " << p.context_ms << ", \"load_ms\": " << p.load_ms << ", \"build_ms\": " << p.build_ms
                           << ", \"render_ms\": " << p.render_ms << ", \"save_ms\": " << p.save_ms << ", \"total_ms\": " << p.total_ms << "}\n";
            }
            if (ti.lastOk && !dump.empty()) {
                std::ofstream f(dump, std::ios::binary);
                f.write((const char*)ti.lastPixels.data(), (std::streamsize)(ti.lastPixels.size() * 12));
                f.write((const char*)ti.lastHitIds.data(), (std::streamsize)(ti.lastHitIds.size() * 4));
            }
            return ti.lastOk ? 0 : 1;   // the library swallows the exception as the reference does; the CLI reports it
        }
    } catch (const std::exception& e) {
        std::fprintf(stderr, "trace_image: %s\n", e.what());
        return 1;
    }
    return usage();
}
