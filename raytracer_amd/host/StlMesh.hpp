// StlMesh.hpp -- from-scratch STL ingest that reproduces the triangle NUMBERING of the reference's vendored
// reader (STLReader.h): hit ids are defined over the post-dedup, post-degenerate-drop triangle list.
//
//   * format sniffing as STLReader.h:729-742: the first 256 bytes, lower-cased, contain "solid", a
//     newline, "facet" and "normal" -> ASCII, else binary;
//   * binary (STLReader.h:666-727): 80-byte header, u32 count, per triangle 12 floats (normal, v0, v1,
//     v2) + 2 attribute bytes;  ASCII (STLReader.h:555-663): tokens, numbers through atof -> float;
//   * RemoveDoubles (STLReader.h:451-535): vertices sorted lexicographically by (x, y, z), exact-equal
//     runs merged, triangles whose three corners are not three distinct vertices are dropped, the
//     others keep file order; normals follow their triangles.
// Not reproduced: which of +0 / -0 survives a merged run (the reference's std::sort leaves it
// unspecified; ties here go to the earliest corner in the file).
#pragma once

#include <algorithm>
#include <cctype>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <sstream>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

namespace hvrt_host {

class StlMeshF {
public:
    StlMeshF() = default;
    explicit StlMeshF(const std::string& filename) { read_file(filename); }
    explicit StlMeshF(const char* filename) { read_file(filename); }

    bool read_file(const std::string& filename) {
        coords_.clear(); normals_.clear(); tris_.clear();
        std::vector<float> corners;     // 3 floats per corner, 3 corners per triangle, file order
        if (looks_ascii(filename)) parse_ascii(filename, corners);
        else parse_binary(filename, corners);
        weld(corners);
        return true;
    }

    size_t num_vrts() const { return coords_.size() / 3; }
    size_t num_tris() const { return tris_.size() / 3; }
    const float* vrt_coords(size_t vi) const { return &coords_[3 * vi]; }
    const unsigned int* tri_corner_inds(size_t ti) const { return &tris_[3 * ti]; }
    const float* tri_normal(size_t ti) const { return &normals_[3 * ti]; }
    const float* raw_coords() const { return coords_.data(); }
    const float* tri_corner_coords(size_t ti, size_t ci) const { return &coords_[3 * tris_[3 * ti + ci]]; }

private:
    std::vector<float> coords_, normals_;
    std::vector<unsigned int> tris_;

    static bool looks_ascii(const std::string& fn) {
        std::ifstream in(fn);
        if (!in) throw std::runtime_error("Couldnt open file " + fn);
        char buf[256];
        in.read(buf, 256);
        std::string s(buf, (size_t)in.gcount());
        std::transform(s.begin(), s.end(), s.begin(), [](unsigned char c) { return (char)std::tolower(c); });
        return s.find("solid") != std::string::npos && s.find("\n") != std::string::npos &&
               s.find("facet") != std::string::npos && s.find("normal") != std::string::npos;
    }

    void parse_binary(const std::string& fn, std::vector<float>& corners) {
        std::ifstream in(fn, std::ios::binary);
        if (!in) throw std::runtime_error("Couldnt open file " + fn);
        char header[80];
        in.read(header, 80);
        if (!in) throw std::runtime_error("Error while parsing binary stl header in file " + fn);
        uint32_t n = 0;
        in.read(reinterpret_cast<char*>(&n), 4);
        if (!in) throw std::runtime_error("Couldnt determine number of triangles in binary stl file " + fn);
        std::vector<char> rec((size_t)n * 50);
        in.read(rec.data(), (std::streamsize)rec.size());
        if ((size_t)in.gcount() != rec.size()) throw std::runtime_error("Error while parsing trianlge in binary stl file " + fn);
        normals_.resize((size_t)n * 3);
        corners.resize((size_t)n * 9);
        for (size_t t = 0; t < n; ++t) {
            std::memcpy(&normals_[3 * t], &rec[50 * t], 12);
            std::memcpy(&corners[9 * t], &rec[50 * t + 12], 36);
        }
    }

    void parse_ascii(const std::string& fn, std::vector<float>& corners) {
        std::ifstream in(fn);
        if (!in) throw std::runtime_error("Couldn't open file " + fn);
        std::string line, tok[8];
        size_t face_vrts = 0, line_no = 0;
        while (std::getline(in, line)) {
            ++line_no;
            std::istringstream ls(line);
            int nt = 0;
            while (nt < 8 && (ls >> tok[nt])) ++nt;
            if (nt == 0) continue;
            auto bad = [&](const char* what) {
                throw std::runtime_error("ERROR while reading from " + fn + ": " + what + " in line " + std::to_string(line_no));
            };
            if (tok[0] == "vertex") {
                if (nt < 4) bad("vertex not specified correctly");
                for (int i = 0; i < 3; ++i) corners.push_back(static_cast<float>(std::atof(tok[i + 1].c_str())));
                ++face_vrts;
            } else if (tok[0] == "facet") {
                if (nt < 5) bad("triangle not specified correctly");
                if (tok[1] != "normal") bad("Missing normal specifier");
                for (int i = 0; i < 3; ++i) normals_.push_back(static_cast<float>(std::atof(tok[i + 2].c_str())));
                face_vrts = 0;
            } else if (tok[0] == "outer") {
                if (nt < 2 || tok[1] != "loop") bad("expecting outer loop");
            } else if (tok[0] == "endfacet") {
                if (face_vrts != 3) bad("bad number of vertices specified for face");
            }
        }
        if (corners.size() / 9 != normals_.size() / 3) throw std::runtime_error("ERROR while reading from " + fn + ": facets and vertices do not pair up");
    }

    // One record per corner: sorted as records (16 bytes, coordinates inline) rather than as indices into the corner
    // array -- the comparisons then read consecutive memory -- and in parallel: chunks sorted by up to 16 threads, merged
    // pairwise.  A 1.0 M-triangle file (3 M corners) took 315 ms in the weld, most of a 4K TraceImage call, with one
    // thread sorting indices.
    struct Corner { float x, y, z; uint32_t i; };
    static bool corner_less(const Corner& a, const Corner& b) {
        if (a.x != b.x) return a.x < b.x;
        if (a.y != b.y) return a.y < b.y;
        if (a.z != b.z) return a.z < b.z;
        return a.i < b.i;
    }
    static void sort_corners(std::vector<Corner>& v) {
        const size_t n = v.size();
        unsigned hw = std::thread::hardware_concurrency();
        if (const char* e = std::getenv("OMP_NUM_THREADS")) { const int k = std::atoi(e); if (k > 0 && (unsigned)k < hw) hw = (unsigned)k; }
        unsigned parts = 1;
        while (parts * 2 <= (hw > 16 ? 16u : hw) && n / (parts * 2) >= 32768) parts *= 2;
        if (parts == 1) { std::sort(v.begin(), v.end(), corner_less); return; }
        auto bound = [&](unsigned k) { return n * k / parts; };
        {
            std::vector<std::thread> th;
            for (unsigned k = 0; k < parts; ++k)
                th.emplace_back([&, k] { std::sort(v.begin() + (std::ptrdiff_t)bound(k), v.begin() + (std::ptrdiff_t)bound(k + 1), corner_less); });
            for (auto& t : th) t.join();
        }
        std::vector<Corner> tmp(n);
        std::vector<Corner>*src = &v, *dst = &tmp;
        for (unsigned w = 1; w < parts; w *= 2) {           // runs of w chunks -> runs of 2 w
            std::vector<std::thread> th;
            for (unsigned k = 0; k < parts; k += 2 * w)
                th.emplace_back([&, k, w] {
                    std::merge(src->begin() + (std::ptrdiff_t)bound(k), src->begin() + (std::ptrdiff_t)bound(k + w),
                               src->begin() + (std::ptrdiff_t)bound(k + w), src->begin() + (std::ptrdiff_t)bound(k + 2 * w),
                               dst->begin() + (std::ptrdiff_t)bound(k), corner_less);
                });
            for (auto& t : th) t.join();
            std::swap(src, dst);
        }
        if (src != &v) v.swap(tmp);
    }

    void weld(const std::vector<float>& c) {
        const size_t nc = c.size() / 3;
        if (nc == 0) { normals_.clear(); return; }
        std::vector<Corner> order(nc);
        for (size_t i = 0; i < nc; ++i) order[i] = { c[3 * i], c[3 * i + 1], c[3 * i + 2], (uint32_t)i };
        sort_corners(order);
        std::vector<uint32_t> remap(nc);
        uint32_t cur = 0;
        coords_.reserve(nc / 2);
        coords_.assign({ order[0].x, order[0].y, order[0].z });
        remap[order[0].i] = 0;
        for (size_t i = 1; i < nc; ++i) {
            const Corner &a = order[i], &b = order[i - 1];
            if (!(a.x == b.x && a.y == b.y && a.z == b.z)) {
                ++cur;
                coords_.push_back(a.x); coords_.push_back(a.y); coords_.push_back(a.z);
            }
            remap[a.i] = cur;
        }
        const size_t nt = nc / 3;
        std::vector<float> kept_normals;
        kept_normals.reserve(normals_.size());
        tris_.reserve(nt * 3);
        for (size_t t = 0; t < nt; ++t) {
            uint32_t a = remap[3 * t], b = remap[3 * t + 1], d = remap[3 * t + 2];
            if (a != b && a != d && b != d) {
                tris_.push_back(a); tris_.push_back(b); tris_.push_back(d);
                kept_normals.insert(kept_normals.end(), &normals_[3 * t], &normals_[3 * t] + 3);
            }
        }
        normals_.swap(kept_normals);
    }
};

}  // namespace hvrt_host

// the reference spells it stl_reader::StlMesh<float, unsigned int>
namespace stl_reader {
template <class TNumber = float, class TIndex = unsigned int>
using StlMesh = ::hvrt_host::StlMeshF;
}
