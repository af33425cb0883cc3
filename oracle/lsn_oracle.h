/*
 * lsn_oracle.h -- CPU restatement of LiveScan3D NativeUtils' fusion hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product:
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library.  The shipped path (livescan3d_amd/) never links or calls it.
 *
 * Pinning status (see DESIGN.md "Oracle"):
 *   - depth -> cloud path (orc_create_vertices / orc_generate_mesh_vertices):
 *     PARITY UNPINNED.  The reference holds no golden vectors for it (ref.bin and
 *     frames_info_*.bin are absent, src/NativeUtils/main.cpp:159-252) and
 *     depthprocessing.cpp cannot be compiled here without stand-ins for
 *     <windows.h> (include/NativeUtils/simpleimage.h:4) and for pgm/simpleimage.
 *   - exact nearest neighbour (orc_nn_*): PINNED against the reference's vendored
 *     nanoflann 1.1.9 + its PointCloud adaptor compiled from /root/reference
 *     (oracle/_ref/ref_nn, fixtures in tests/golden/nn_*.npz).
 *   - triangulation (orc_generate_triangles): PINNED against the reference's own
 *     src/NativeUtils/meshGenerator.cpp compiled where it lies (oracle/_ref/libref_tri.so,
 *     fixtures in tests/golden/tri_reference.npz).
 *   - radial correction: PARITY UNPINNED (lives in depthprocessing.cpp, see above).
 *   - rest of ICP (matching, rejection, Kabsch): PARITY UNPINNED (needs OpenCV
 *     3.2.0 core binaries, absent; the only reference ICP test is commented out,
 *     src/NativeUtils/main.cpp:253-268).
 *
 * Every function cites the reference file:line it follows.  Paths are relative
 * to the reference repository root.
 */
#ifndef LSN_ORACLE_H
#define LSN_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* include/NativeUtils/depthprocessing.h:29-33 */
typedef struct { unsigned char R, G, B, A; float X, Y, Z; } orc_vertex;

/* createVertices, src/NativeUtils/depthprocessing.cpp:122-187 (+ RotatePoint :109-120)
 * followed by the AoS repack of formMesh :1594-1608 for one sensor.
 * intr7 = {cx,cy,fx,fy,r2,r4,r6}; wt12 = {t[3], R[3][3] row-major}; bounds6 = {minX,minY,minZ,maxX,maxY,maxZ}.
 * out must hold w*h vertices.  vert_to_pix (nullable) receives vertices_to_depth_map,
 * pix_to_vert (nullable, w*h ints) receives depth_to_vertices_map (-1 = none).
 * Returns the number of vertices. */
int orc_create_vertices(const uint16_t *depth, const uint8_t *rgb, int w, int h,
                        const float *intr7, const float *wt12, const float *bounds6,
                        orc_vertex *out, int *vert_to_pix, int *pix_to_vert);

/* generateMeshFromDepthMaps with (bcolor_transfer,bgenerate_triangles)=(false,false), vertices only:
 * src/NativeUtils/depthprocessing.cpp:1715-1792 -> generateVerticesFromDepthMaps :708-733 -> formMesh :1578-1608.
 * depth_maps / depth_colors are the concatenated per-sensor buffers (:1646-1650 striding).
 * out must hold sum(w*h) vertices; per_map_counts (nullable) gets n_maps counts.  Returns nVertices.
 * n_threads > 1 runs one sensor per thread like the reference's std::thread fan-out (P1). */
long orc_generate_mesh_vertices(int n_maps, const uint8_t *depth_maps, const uint8_t *depth_colors,
                                const int *widths, const int *heights,
                                const float *intr, const float *wt, const float *bounds6,
                                orc_vertex *out, int *per_map_counts, int n_threads);

/* generateVerticesFromDepthMap, src/NativeUtils/depthprocessing.cpp:1631-1657. */
int orc_generate_vertices_from_depth_map(const uint8_t *depth_maps, const uint8_t *depth_colors,
                                         const int *widths, const int *heights,
                                         const float *intr, const float *wt, const float *bounds6,
                                         int depth_map_index, orc_vertex *out);

/* depthMapAndColorRadialCorrection, src/NativeUtils/depthprocessing.cpp:191-261: forward warp of depth + colour with
 * d = 1 - r2 r - r4 r^2 - r6 r^3 (later source pixels overwrite earlier ones), then the in-place, raster-order 8-neighbour
 * hole closing (> 4 neighbours within 30 mm of the previously accepted one).  In place on depth (w*h u16) and rgb (w*h*3).
 * The (int) casts follow x86-64 cvttss2si: NaN / out-of-range -> INT_MIN (then rejected by the >= 0 test).
 * PARITY UNPINNED (same reason as the depth path). */
void orc_radial_correction(uint16_t *depth, uint8_t *rgb, int w, int h, const float *intr7);

/* depthMapAndColorSetRadialCorrection (export), depthprocessing.cpp:1794-1815: every sensor of the concatenated buffers. */
void orc_radial_correction_all(int n_maps, uint8_t *depth_maps, uint8_t *depth_colors, const int *widths, const int *heights,
                               const float *intr, int n_threads);

/* MeshGenerator::generateTrianglesGradients, src/NativeUtils/meshGenerator.cpp:14-181 (driver
 * generateTriangles, src/NativeUtils/depthprocessing.cpp:1659-1691): 2x2-stencil triangulation of ONE sensor.
 * depth = the sensor's depth map (VerticesWithDepthColorMaps::depth_map, a copy of the input, depthprocessing.cpp:180),
 * pix_to_vert = depth_to_vertices_map (-1 = no vertex).  The reference splits the rows over 4 threads and concatenates
 * the bands in order, which is plain raster order over y in [2,h-2), x in [1,w-2).  index_base is added to every
 * index (formMesh rebases by the cumulative vertex count, depthprocessing.cpp:1611-1627).
 * out must hold 2*w*h triangles (3 ints each).  Returns the number of triangles.
 * PINNED: bit-exact against meshGenerator.cpp itself (oracle/_ref/libref_tri.so, tests/golden/tri_reference.npz). */
long orc_generate_triangles(const uint16_t *depth, const int *pix_to_vert, int w, int h, int index_base, int *out);

/* generateMeshFromDepthMaps with flags (false,false), complete: vertices (as orc_generate_mesh_vertices) AND the
 * always-on triangulation (depthprocessing.cpp:1786) with rebased indices.  out_tri must hold 2*sum(w*h)*3 ints.
 * Returns nVertices; *n_triangles receives nTriangles. */
long orc_generate_mesh(int n_maps, const uint8_t *depth_maps, const uint8_t *depth_colors,
                       const int *widths, const int *heights, const float *intr, const float *wt,
                       const float *bounds6, orc_vertex *out, int *per_map_counts, int *out_tri, long *n_triangles);

/* Exact 1-NN, squared L2 evaluated as (d0*d0 + d1*d1) + d2*d2 in f32
 * (include/NativeUtils/icp.h:40-47; query loop src/NativeUtils/icp.cpp:18-32).
 * Ties (equal f32 distance) resolve to the LOWEST target index -- nanoflann's tie
 * order depends on its traversal; fixtures are tie-free.
 * targets: n1*3 floats, queries: n2*3 floats. */
void orc_nn_brute(const float *targets, int n1, const float *queries, int n2,
                  int64_t *idx, float *dist, int n_threads);
void orc_nn_kdtree(const float *targets, int n1, const float *queries, int n2,
                   int64_t *idx, float *dist, int n_threads);

/* Per-iteration trace of orc_icp (all optional, for piecewise tests). */
typedef struct {
    int   n_matched;      /* m  : one-to-one matches before rejection (icp.cpp:95-126) */
    int   n_kept;         /* m' : after RejectOutlierMatches (icp.cpp:56-73)          */
    float mean, stddev;   /* GetStandardDeviation (icp.cpp:34-54)                      */
    float T[3];           /* tempT (icp.cpp:141)                                       */
    float Rn[9];          /* tempR (icp.cpp:155-163), row-major                        */
} orc_icp_iter;

/* ICP, src/NativeUtils/icp.cpp:75-177.  verts2 is moved in place; R (9, row-major) and t (3)
 * are in/out.  nn_mode 0 = brute force, 1 = kd-tree (same results).  trace (nullable) must
 * hold maxIter entries.  Returns 1.0f like the reference (:86,:176). */
float orc_icp(const float *verts1, float *verts2, int n1, int n2,
              float *R, float *t, int maxIter, int nn_mode, int n_threads,
              orc_icp_iter *trace);

/* refineWorker_DoWork, LiveScanServer/MainWindowForm.cs:330-410: Gauss-Seidel ICP over sensors and
 * the (aliasing) pose composition.  clouds[i] = n[i]*3 floats, moved in place.
 * world_R (n_sensors*9) / world_t (n_sensors*3) are updated like worldTransforms[i]. */
void orc_refine(int n_sensors, float **clouds, const int *n, int n_refine_iters, int n_icp_iters,
                float *world_R, float *world_t, float *Rs_out, float *Ts_out,
                int nn_mode, int n_threads);

/* 3x3 SVD-based rotation: Rn = U*Vt (last column of U negated if det<0), icp.cpp:152-163. */
void orc_kabsch_rotation(const float *M9, float *Rn9);

/* ---- wire / disk formats either side of the path.  PARITY UNPINNED: the C# side cannot run here
 * (no dotnet/mono) and the reference holds no sample files; these follow the cited lines literally. ---- */

#define ORC_CHUNK_LIMIT (65000 - 3)   /* TransferServer.cs:179,205 */

/* TransferServer.formMeshChunks (LiveScanServer/TransferServer.cs:203-270) when nt > 0, else formVerticesChunks
 * (:177-201).  new_v must hold max(nv, 3*nt) vertices, new_tri 3*nt ints, v_chunks / t_chunks 3*nt/ORC_CHUNK_LIMIT + nv/ORC_CHUNK_LIMIT + 2
 * ints.  Returns the number of chunks; *n_new_v receives the number of vertices to send.
 * The triangle count of the first chunk is one short when there is more than one chunk (trianglesChunkStart = t at
 * :244 instead of t+1) -- reproduced as written. */
int orc_form_chunks(const orc_vertex *v, int nv, const int *tri, int nt, orc_vertex *new_v, int *new_tri,
                    int *v_chunks, int *t_chunks, int *n_new_v);

/* The bytes TransferSocket.SendFrame writes (LiveScanServer/TransferSocket.cs:50-104) for the cloud/mesh the
 * TransferServer would send (TransferServer.cs:142-157): int nVertices, int nTriangles, int nChunks,
 * int vChunkSizes[nChunks], int tChunkSizes[nChunks], float xyz[3*nVertices], u8 rgb[3*nVertices], int tri[3*nTriangles].
 * Returns the length, or -(needed) when cap is too small. */
long orc_transfer_frame(const orc_vertex *v, int nv, const int *tri, int nt, uint8_t *out, long cap);

/* Utils.saveToPly(..., binary = true) file bytes (LiveScanServer/Utils.cs:222-262): header (first line ends with the
 * Windows StreamWriter.WriteLine "\r\n", every other line with "\n"), 15-byte vertex records {f32 x,y,z; u8 r,g,b},
 * 13-byte face records {u8 3; i32 a,b,c}. */
long orc_ply_binary(const orc_vertex *v, int nv, const int *tri, int nt, uint8_t *out, long cap);

/* Frame message client -> server: LiveScanClient::SerializeFrame after the colour mapping
 * (src/LiveScanClient/liveScanClient.cpp:209-290), read by KinectSocket.ReceiveFrame (LiveScanServer/KinectSocket.cs:211-304):
 * 16-byte header {i32 payload bytes, i32 compressed, i32 w, i32 h} + payload {u16 depth[w*h], u8 rgb[3*w*h], bodies}.
 * The oracle writes/reads the uncompressed form only (compressed = 0).  bodies = the serialized body block
 * (i32 nBodies, then per body u8 tracked, i32 nJoints, nJoints x 28 bytes), at least the 4 bytes of nBodies. */
long orc_frame_encode(const uint8_t *depth, const uint8_t *rgb, int w, int h, const uint8_t *bodies, int bodies_bytes,
                      uint8_t *out, long cap);
/* Returns 0 and pointers into msg, or -1 when the message is malformed (short, compressed, inconsistent body block). */
int orc_frame_decode(const uint8_t *msg, long len, int *w, int *h, const uint8_t **depth, const uint8_t **rgb,
                     const uint8_t **bodies, int *bodies_bytes, int *n_bodies);

/* Recording file of the client (src/LiveScanClient/frameFileWriterReader.cpp:59-82 reader, :115-130 writer): records
 * "bufferSize= %d\nframe_timestamp= %d\n" + bytes + "\n".  append returns the bytes written (or -(needed));
 * next returns the position after the record (or -1 at the end / on a malformed record). */
long orc_recording_append(uint8_t *out, long cap, const uint8_t *frame, int len, int timestamp_ms);
long orc_recording_next(const uint8_t *file, long len, long pos, long *frame_off, int *frame_len, int *timestamp_ms);

#ifdef __cplusplus
}
#endif
#endif
