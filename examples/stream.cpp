// stream.cpp -- one LiveScanServer tick loop fed from client recordings, as a plain C++ host of libNativeUtils.so:
//   inbound : one recording file per client (src/LiveScanClient/frameFileWriterReader.cpp:115-130), each record a frame
//             message (src/LiveScanClient/liveScanClient.cpp:185-290) that KinectSocket.ReceiveFrame would read
//             (LiveScanServer/KinectSocket.cs:211-304): lsnRecordingNext + lsnFrameParseHeader + lsnFrameDecode;
//   path    : KinectServer.GenerateMesh (LiveScanServer/KinectServer.cs:354-374) = generateMeshFromDepthMaps;
//   outbound: what TransferServer would put on the socket for that mesh (TransferServer.cs:142-157 + TransferSocket.cs:50-104)
//             = lsnLastMeshTransferFrame, appended tick after tick to --frames-out; the last tick's mesh as a binary PLY
//             (Utils.saveToPly, LiveScanServer/Utils.cs:222-262) = lsnLastMeshPly, written to --ply.
//
//   stream --calib calib.bin [--bounds 6 floats] [--frames-out f.bin] [--ply mesh.ply] rec0.bin rec1.bin ...
//   calib.bin: per sensor 7 f32 intrinsics + 12 f32 pose (the arrays KinectServer passes, KinectServer.cs:470-490)
//
// Build: make -C examples stream   (links -lNativeUtils only)
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
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
    bool ok = n == 0 || fread(buf.data(), 1, (size_t)n, f) == (size_t)n;
    fclose(f);
    return ok;
}

static int fail(const char *what)
{
    char err[512] = "";
    lsnGetLastError(err, sizeof err);
    fprintf(stderr, "%s: %s\n", what, err);
    return 1;
}

int main(int argc, char **argv)
{
    float b[6] = {-5, -5, -5, 5, 5, 5};   // KinectSettings.cs:54-60
    const char *calib = nullptr, *frames_out = nullptr, *ply = nullptr;
    std::vector<const char *> recs;
    for (int i = 1; i < argc; i++) {
        if (!strcmp(argv[i], "--calib") && i + 1 < argc) calib = argv[++i];
        else if (!strcmp(argv[i], "--frames-out") && i + 1 < argc) frames_out = argv[++i];
        else if (!strcmp(argv[i], "--ply") && i + 1 < argc) ply = argv[++i];
        else if (!strcmp(argv[i], "--bounds") && i + 6 < argc) { for (int k = 0; k < 6; k++) b[k] = (float)atof(argv[++i]); }
        else recs.push_back(argv[i]);
    }
    const int n = (int)recs.size();
    if (!calib || n == 0) { fprintf(stderr, "usage: %s --calib calib.bin [--bounds 6 floats] [--frames-out f.bin] [--ply mesh.ply] rec0.bin ...\n", argv[0]); return 2; }
    std::vector<unsigned char> cal;
    if (!read_all(calib, cal) || cal.size() != (size_t)n * 19 * 4) { fprintf(stderr, "calib: expected %d x 19 floats\n", n); return 1; }
    std::vector<float> intr(7 * (size_t)n), wt(12 * (size_t)n);
    for (int i = 0; i < n; i++) {
        memcpy(&intr[7 * i], cal.data() + 76 * (size_t)i, 28);
        memcpy(&wt[12 * i], cal.data() + 76 * (size_t)i + 28, 48);
    }
    std::vector<std::vector<unsigned char>> files(n);
    std::vector<long long> pos(n, 0);
    for (int i = 0; i < n; i++)
        if (!read_all(recs[i], files[i])) return 1;

    FILE *fo = frames_out ? fopen(frames_out, "wb") : nullptr;
    if (frames_out && !fo) { perror(frames_out); return 1; }
    std::vector<int> w(n), h(n);
    std::vector<unsigned char> depth, color, wire;
    int ticks = 0;
    long long sent = 0;
    for (;; ticks++) {
        depth.clear();
        color.clear();
        bool end = false;
        for (int i = 0; i < n && !end; i++) {
            long long off = 0;
            int len = 0, ts = 0;
            const long long next = lsnRecordingNext(files[i].data(), (long long)files[i].size(), pos[i], &off, &len, &ts);
            if (next < 0) { end = true; break; }                       // a recording ran out: the session is over
            pos[i] = next;
            LsnFrameInfo info;
            if (len < 16 || lsnFrameParseHeader(files[i].data() + off, &info) != 0 || 16 + (long long)info.payload_bytes > len) { end = true; break; }
            w[i] = info.width;
            h[i] = info.height;
            const size_t npx = (size_t)info.width * info.height, d0 = depth.size(), c0 = color.size();
            depth.resize(d0 + 2 * npx);
            color.resize(c0 + 3 * npx);
            if (lsnFrameDecode(files[i].data() + off + 16, info.payload_bytes, info.compressed, info.width, info.height,
                               depth.data() + d0, color.data() + c0, nullptr, 0, nullptr) < 0)
                return fail("lsnFrameDecode");
        }
        if (end) break;
        Mesh mesh;
        memset(&mesh, 0, sizeof mesh);
        generateMeshFromDepthMaps(n, depth.data(), color.data(), w.data(), h.data(), intr.data(), wt.data(), &mesh, false,
                                  b[0], b[1], b[2], b[3], b[4], b[5], false);
        const int nv = mesh.nVertices, nt = mesh.nTriangles;
        deleteMesh(&mesh);
        if (nv == 0) { char e[8]; if (lsnGetLastError(e, sizeof e) > 0) return fail("generateMeshFromDepthMaps"); }
        const long long bound = lsnLastMeshTransferFrame(nullptr, 0);
        if (bound < 0) return fail("lsnLastMeshTransferFrame");
        wire.resize((size_t)bound);
        const long long len = lsnLastMeshTransferFrame(wire.data(), bound);
        if (len < 0) return fail("lsnLastMeshTransferFrame");
        if (fo) fwrite(wire.data(), 1, (size_t)len, fo);
        sent += len;
        printf("tick %d: %d vertices, %d triangles -> %lld bytes on the wire\n", ticks, nv, nt, len);
    }
    if (fo) fclose(fo);
    if (ply && ticks > 0) {
        const long long need = lsnLastMeshPly(nullptr, 0);
        if (need < 0) return fail("lsnLastMeshPly");
        wire.resize((size_t)need);
        if (lsnLastMeshPly(wire.data(), need) != need) return fail("lsnLastMeshPly");
        FILE *fp = fopen(ply, "wb");
        if (!fp) { perror(ply); return 1; }
        fwrite(wire.data(), 1, (size_t)need, fp);
        fclose(fp);
    }
    printf("%d ticks, %lld bytes streamed\n", ticks, sent);
    return 0;
}
