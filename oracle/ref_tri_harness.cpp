// ref_tri_harness.cpp -- drives the REFERENCE's own triangulation.
//
// TEST INFRASTRUCTURE ONLY.  Compiled by oracle/Makefile (target _ref/libref_tri.so) together with
// src/NativeUtils/meshGenerator.cpp where it lies under /root/reference; no reference source is copied
// into this repository.  Recipe-level accommodations, all on the compiler command line: -D__int64='long long'
// (MSVC keyword at meshGenerator.cpp:39) and -include cstring (memcpy at :178, which MSVC's <vector> drags in).
// The call below is the one depthprocessing.cpp:844 / :1677 makes per sensor.
#include "NativeUtils/meshGenerator.h"
#include <cstddef>
#include <cstdint>
#include <cstring>

// Mesh::triangles is `int*`, three ints per triangle, as formMesh copies TriangleIndexes out (depthprocessing.cpp:1611-1627): the
// reference's own struct must be exactly that, or the memcpy below (and this library's int[3] layout) would be wrong.
static_assert(sizeof(TriangleIndexes) == 3 * sizeof(int) && offsetof(TriangleIndexes, ind) == 0, "TriangleIndexes is three ints");

// depth: w*h u16 (the sensor's depth map as createVertices copies it, depthprocessing.cpp:181);
// pix_to_vert: w*h ints, -1 = no vertex.  out must hold 2*w*h*3 ints.  Returns the triangle count.
extern "C" long ref_generate_triangles(const uint16_t *depth, const int *pix_to_vert, int w, int h, int *out)
{
    std::vector<UINT16> d(depth, depth + (size_t)w * h);
    std::vector<int> map(pix_to_vert, pix_to_vert + (size_t)w * h);
    std::vector<TriangleIndexes> tri;
    MeshGenerator::generateTrianglesGradients(d.data(), map, tri, w, h);
    if (!tri.empty()) memcpy(out, tri.data(), tri.size() * sizeof(TriangleIndexes));
    return (long)tri.size();
}
