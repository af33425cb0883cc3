// ref_nn_harness.cpp -- drives the REFERENCE's own nearest-neighbour step.
//
// TEST INFRASTRUCTURE ONLY.  Compiled by oracle/Makefile (target _ref/ref_nn) against the headers
// where they lie under /root/reference: include/NativeUtils/icp.h (PointCloud adaptor, :33-62) and the
// vendored include/nanoflann.h (v1.1.9).  No reference source is copied into this repository; the
// only recipe-level accommodation is -D__declspec(x)= -D__stdcall= for the MSVC keywords in icp.h:7-10,65.
// The loop below is what FindClosestPointForEach does (src/NativeUtils/icp.cpp:18-32) minus cv::Mat,
// which is only a row-pointer carrier there.
//
// Usage: ref_nn <in.bin> <out.bin>
//   in : int32 n1, int32 n2, float32 targets[n1*3], float32 queries[n2*3]
//   out: int64 idx[n2], float32 dist[n2]
// Also built as _ref/libref_nn.so exporting ref_nn_query() for timing (bench.py cpu_baseline).
#include "NativeUtils/icp.h"
#include <cstdio>
#include <cstdlib>
#include <cstdint>

extern "C" void ref_nn_query(const float *targets, int n1, const float *queries, int n2,
                             int64_t *idx, float *dist)
{
    PointCloud cloud;
    cloud.pts.resize(n1);
    for (int i = 0; i < n1; i++) { cloud.pts[i].X = targets[3*i]; cloud.pts[i].Y = targets[3*i+1]; cloud.pts[i].Z = targets[3*i+2]; }
    typedef nanoflann::KDTreeSingleIndexAdaptor<nanoflann::L2_Simple_Adaptor<float, PointCloud>, PointCloud, 3> kdTree;
    kdTree tree(3, cloud);
    tree.buildIndex();
    std::vector<size_t> indices(n2);
    std::vector<float> distances(n2);
#pragma omp parallel for
    for (int i = 0; i < n2; i++) {
        nanoflann::KNNResultSet<float> resultSet(1);
        resultSet.init(&indices[i], &distances[i]);
        tree.findNeighbors(resultSet, queries + 3 * (size_t)i, nanoflann::SearchParams());
    }
    for (int i = 0; i < n2; i++) { idx[i] = (int64_t)indices[i]; dist[i] = distances[i]; }
}

#ifndef REF_NN_NO_MAIN
int main(int argc, char **argv)
{
    if (argc != 3) { fprintf(stderr, "usage: %s in.bin out.bin\n", argv[0]); return 2; }
    FILE *f = fopen(argv[1], "rb");
    if (!f) { perror(argv[1]); return 1; }
    int32_t n1, n2;
    if (fread(&n1, 4, 1, f) != 1 || fread(&n2, 4, 1, f) != 1) return 1;
    std::vector<float> t((size_t)n1 * 3), q((size_t)n2 * 3);
    if (fread(t.data(), 4, t.size(), f) != t.size() || fread(q.data(), 4, q.size(), f) != q.size()) return 1;
    fclose(f);
    std::vector<int64_t> idx(n2);
    std::vector<float> dist(n2);
    ref_nn_query(t.data(), n1, q.data(), n2, idx.data(), dist.data());
    f = fopen(argv[2], "wb");
    if (!f) { perror(argv[2]); return 1; }
    fwrite(idx.data(), 8, idx.size(), f);
    fwrite(dist.data(), 4, dist.size(), f);
    fclose(f);
    return 0;
}
#endif
