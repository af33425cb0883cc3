// shard.cpp -- one rank of the multi-GPU fusion step as a plain C++ host of libNativeUtils.so: no HIP, no RCCL, no MPI and no
// Python on this side of the C-ABI.  Every rank (one process per GPU) reads the same capture file (the reference's own format,
// src/NativeUtils/depthprocessing.cpp:1316-1385), uploads ITS block of sensors [rank * n / world, (rank + 1) * n / world),
// and lsnShardStep returns the merged cloud of ALL sensors in formMesh's sensor order (depthprocessing.cpp:1594-1608) --
// what the reference's per-sensor std::thread fan-out (:708-733) and concatenation do inside one process.
// Rendezvous: rank 0 writes the 128 bytes of lsnShardUniqueId to --id-file, the other ranks wait for the file.
//
//   shard <frames.bin> --rank R --world W --id-file PATH [--device D] [--bounds 6 floats] [--expect mesh.bin]
//
// --expect compares the vertices with a golden mesh file (the triangle part of the file is skipped: the step fuses vertices).
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "../include/NativeUtils.h"

static bool read_all(const char *path, std::vector<unsigned char> &buf)
{
    FILE *f = fopen(path, "rb");
    if (!f) return false;
    fseek(f, 0, SEEK_END);
    long n = ftell(f);
    fseek(f, 0, SEEK_SET);
    buf.resize((size_t)n);
    bool ok = fread(buf.data(), 1, (size_t)n, f) == (size_t)n;
    fclose(f);
    return ok;
}

static int fail(const char *what)
{
    char err[512];
    lsnGetLastError(err, sizeof(err));
    fprintf(stderr, "%s: %s\n", what, err);
    return 1;
}

int main(int argc, char **argv)
{
    if (argc < 2) { fprintf(stderr, "usage: %s frames.bin --rank R --world W --id-file PATH [--device D] [--bounds 6 floats] [--expect mesh.bin]\n", argv[0]); return 2; }
    float b[6] = {-5, -5, -5, 5, 5, 5};
    int rank = 0, world = 1, device = -1;
    const char *id_file = nullptr, *expect = nullptr;
    for (int i = 2; i < argc; i++) {
        if (!strcmp(argv[i], "--rank") && i + 1 < argc) rank = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--world") && i + 1 < argc) world = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--device") && i + 1 < argc) device = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--id-file") && i + 1 < argc) id_file = argv[++i];
        else if (!strcmp(argv[i], "--expect") && i + 1 < argc) expect = argv[++i];
        else if (!strcmp(argv[i], "--bounds") && i + 6 < argc) { for (int k = 0; k < 6; k++) b[k] = (float)atof(argv[++i]); }
    }
    if (device < 0) device = rank;
    if (!id_file && world > 1) { fprintf(stderr, "--id-file is needed with more than one rank\n"); return 2; }

    std::vector<unsigned char> raw;
    if (!read_all(argv[1], raw) || raw.size() < 4) { perror(argv[1]); return 1; }
    size_t pos = 0;
    int n;
    memcpy(&n, raw.data(), 4); pos = 4;
    std::vector<int> w(n), h(n);
    memcpy(w.data(), raw.data() + pos, 4 * (size_t)n); pos += 4 * (size_t)n;
    memcpy(h.data(), raw.data() + pos, 4 * (size_t)n); pos += 4 * (size_t)n;
    if (n % world != 0) { fprintf(stderr, "%d sensors do not split evenly over %d ranks\n", n, world); return 2; }
    const int per = n / world, first = rank * per;
    std::vector<unsigned char> depth, color;   // this rank's block only
    for (int i = 0; i < n; i++) {
        const size_t npx = (size_t)w[i] * h[i];
        if (i >= first && i < first + per) {
            depth.insert(depth.end(), raw.begin() + pos, raw.begin() + pos + 2 * npx);
            color.insert(color.end(), raw.begin() + pos + 2 * npx, raw.begin() + pos + 5 * npx);
        }
        pos += 5 * npx;
    }
    std::vector<float> intr(7 * (size_t)n), wt(12 * (size_t)n);
    memcpy(intr.data(), raw.data() + pos, 28 * (size_t)n); pos += 28 * (size_t)n;
    memcpy(wt.data(), raw.data() + pos, 48 * (size_t)n); pos += 48 * (size_t)n;

    // rendezvous: 128 bytes from rank 0
    unsigned char id[128];
    if (rank == 0) {
        if (lsnShardUniqueId(id)) return fail("lsnShardUniqueId");
        if (id_file) {
            std::string tmp = std::string(id_file) + ".tmp";
            FILE *f = fopen(tmp.c_str(), "wb");
            if (!f || fwrite(id, 1, 128, f) != 128) { perror(id_file); return 1; }
            fclose(f);
            rename(tmp.c_str(), id_file);   // appears atomically
        }
    } else {
        std::vector<unsigned char> got;
        for (int tries = 0; tries < 600 && (!read_all(id_file, got) || got.size() != 128); tries++) std::this_thread::sleep_for(std::chrono::milliseconds(100));
        if (got.size() != 128) { fprintf(stderr, "rank %d: no id in %s\n", rank, id_file); return 1; }
        memcpy(id, got.data(), 128);
    }

    LsnShard *sh = lsnShardCreate(device, rank, world, id, 1, n, w.data(), h.data());   // collective
    if (!sh) return fail("lsnShardCreate");
    void *stream = lsnStreamCreate(device);
    void *d_depth = lsnDeviceMalloc(device, (long long)depth.size() + 64), *d_color = lsnDeviceMalloc(device, (long long)color.size() + 64);
    if (!stream || !d_depth || !d_color) return fail("device memory");
    if (lsnDeviceUpload(device, d_depth, depth.data(), (long long)depth.size(), stream) ||
        lsnDeviceUpload(device, d_color, color.data(), (long long)color.size(), stream) ||
        lsnShardSetParams(sh, intr.data(), wt.data(), b, stream))
        return fail("upload");
    void *d_merged = nullptr;
    int *d_off = nullptr;
    auto t0 = std::chrono::steady_clock::now();
    if (lsnShardStep(sh, d_depth, d_color, &d_merged, &d_off, stream)) return fail("lsnShardStep");
    std::vector<int> off(n + 1);
    if (lsnDeviceDownload(device, off.data(), d_off, 4ll * (n + 1), stream) || lsnStreamSynchronize(device, stream)) return fail("download");
    const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    const int nv = off[n];
    std::vector<VertexC4ubV3f> verts((size_t)nv);
    if (nv > 0 && (lsnDeviceDownload(device, verts.data(), d_merged, 16ll * nv, stream) || lsnStreamSynchronize(device, stream))) return fail("download");
    printf("rank %d of %d: sensors [%d, %d) in, merged cloud of %d sensors out: %d vertices, first step %.3f ms, %lld bytes sent\n", rank, world, first,
           first + per, n, nv, ms, lsnShardLastBytesSent(sh));
    int rc = 0;
    if (expect) {
        std::vector<unsigned char> g;
        if (!read_all(expect, g)) { perror(expect); return 1; }
        int nt, gnv;
        memcpy(&nt, g.data(), 4);
        memcpy(&gnv, g.data() + 4 + 12 * (size_t)nt, 4);
        const unsigned char *gv = g.data() + 8 + 12 * (size_t)nt;
        if (gnv != nv) { printf("Numbers of vertices are not equal! (%d expected)\n", gnv); rc = 3; }
        else if (memcmp(gv, verts.data(), 16 * (size_t)nv) != 0) { printf("Different vertex!\n"); rc = 3; }
        else printf("Test PASSED\n");
    }
    lsnDeviceFree(device, d_depth);
    lsnDeviceFree(device, d_color);
    lsnStreamDestroy(device, stream);
    lsnShardDestroy(sh);
    return rc;
}
