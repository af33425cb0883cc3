// replay.cpp -- what LiveScanServer does around the native merge call, as a plain C++ host of libNativeUtils.so:
// KinectServer.CopyLatestFrames' argument packing (LiveScanServer/KinectServer.cs:453-498, here read from the reference's
// own capture file, src/NativeUtils/depthprocessing.cpp:1316-1385), CorrectRadialDistortionsForDepthMaps (:518-525,
// optional), GenerateMesh (:354-374) and the mesh copy-out, then the bit-for-bit comparison of the reference's
// regression main() (src/NativeUtils/main.cpp:211-245) against a golden mesh file.
//
//   replay <frames.bin> [--radial] [--bounds minX minY minZ maxX maxY maxZ] [--expect mesh.bin] [--reps N]
//
// Build: make -C examples      (links -lNativeUtils only; no HIP or torch types on this side of the boundary)
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../include/NativeUtils.h"

static bool read_all(const char *path, std::vector<unsigned char> &buf)
{
    FILE *f = fopen(path, "rb");
    if (!f) { perror(path); return false; }
    fseek(f, 0, SEEK_END);
    long n = ftell(f);
    fseek(f, 0, SEEK_SET);
    buf.resize((size_t)n);
    bool ok = fread(buf.data(), 1, (size_t)n, f) == (size_t)n;
    fclose(f);
    return ok;
}

int main(int argc, char **argv)
{
    if (argc < 2) { fprintf(stderr, "usage: %s frames.bin [--radial] [--bounds 6 floats] [--expect mesh.bin] [--reps N]\n", argv[0]); return 2; }
    float b[6] = {-5, -5, -5, 5, 5, 5};   // KinectSettings.cs:54-60
    const char *expect = nullptr;
    bool radial = false;
    int reps = 1;
    for (int i = 2; i < argc; i++) {
        if (!strcmp(argv[i], "--radial")) radial = true;
        else if (!strcmp(argv[i], "--bounds") && i + 6 < argc) { for (int k = 0; k < 6; k++) b[k] = (float)atof(argv[++i]); }
        else if (!strcmp(argv[i], "--expect") && i + 1 < argc) expect = argv[++i];
        else if (!strcmp(argv[i], "--reps") && i + 1 < argc) reps = atoi(argv[++i]);
    }
    std::vector<unsigned char> raw;
    if (!read_all(argv[1], raw) || raw.size() < 4) return 1;
    size_t pos = 0;
    int n;
    memcpy(&n, raw.data(), 4); pos = 4;
    std::vector<int> w(n), h(n);
    memcpy(w.data(), raw.data() + pos, 4 * (size_t)n); pos += 4 * (size_t)n;
    memcpy(h.data(), raw.data() + pos, 4 * (size_t)n); pos += 4 * (size_t)n;
    std::vector<unsigned char> depth, color;
    for (int i = 0; i < n; i++) {
        size_t npx = (size_t)w[i] * h[i];
        depth.insert(depth.end(), raw.begin() + pos, raw.begin() + pos + 2 * npx); pos += 2 * npx;
        color.insert(color.end(), raw.begin() + pos, raw.begin() + pos + 3 * npx); pos += 3 * npx;
    }
    std::vector<float> intr(7 * (size_t)n), wt(12 * (size_t)n);
    memcpy(intr.data(), raw.data() + pos, 28 * (size_t)n); pos += 28 * (size_t)n;
    memcpy(wt.data(), raw.data() + pos, 48 * (size_t)n); pos += 48 * (size_t)n;
    if (pos != raw.size()) { fprintf(stderr, "frames file: %zu trailing bytes\n", raw.size() - pos); return 1; }

    Mesh mesh;
    memset(&mesh, 0, sizeof(mesh));
    std::vector<VertexC4ubV3f> verts;
    std::vector<int> tris;
    double best_ms = 1e30;
    for (int r = 0; r < reps; r++) {
        std::vector<unsigned char> d = depth, c = color;   // radial correction works in place
        auto t0 = std::chrono::steady_clock::now();
        if (radial) depthMapAndColorSetRadialCorrection(n, d.data(), c.data(), w.data(), h.data(), intr.data());
        generateMeshFromDepthMaps(n, d.data(), c.data(), w.data(), h.data(), intr.data(), wt.data(), &mesh, false,
                                  b[0], b[1], b[2], b[3], b[4], b[5], false);
        // CopyMeshToVerticesWithColoursArray / CopyMeshToTrianglesArray (KinectServer.cs:342-389), then deleteMesh
        verts.assign(mesh.vertices, mesh.vertices + mesh.nVertices);
        tris.assign(mesh.triangles, mesh.triangles + 3 * (size_t)mesh.nTriangles);
        deleteMesh(&mesh);
        double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        if (ms < best_ms) best_ms = ms;
    }
    char err[512];
    if (lsnGetLastError(err, sizeof(err)) > 0) fprintf(stderr, "NativeUtils: %s\n", err);
    printf("sensors %d  vertices %zu  triangles %zu  best %.3f ms/call\n", n, verts.size(), tris.size() / 3, best_ms);

    if (expect) {
        std::vector<unsigned char> g;
        if (!read_all(expect, g)) return 1;
        int nt, nv;
        memcpy(&nt, g.data(), 4);
        const int *gt = reinterpret_cast<const int *>(g.data() + 4);
        memcpy(&nv, g.data() + 4 + 12 * (size_t)nt, 4);
        const unsigned char *gv = g.data() + 8 + 12 * (size_t)nt;
        if ((size_t)nv != verts.size()) { printf("Numbers of vertices are not equal! (%d expected)\n", nv); return 3; }
        if ((size_t)nt != tris.size() / 3) { printf("Numbers of triangles are not equal! (%d expected)\n", nt); return 3; }
        if (memcmp(gt, tris.data(), 12 * (size_t)nt) != 0) { printf("Different triangle!\n"); return 3; }
        if (memcmp(gv, verts.data(), 16 * (size_t)nv) != 0) { printf("Different vertex!\n"); return 3; }
        printf("Test PASSED\n");
    }
    return 0;
}
