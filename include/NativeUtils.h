/*
 * NativeUtils.h -- C-ABI of libNativeUtils.so, the MI355X-native drop-in for LiveScan3D's NativeUtils.dll
 * on the per-tick fusion path (depth -> XYZ unprojection, R(p+t), AABB crop, raster-order merged cloud, ICP).
 *
 * Part 1 are the reference's own exports, same names, argument order and meaning, so LiveScanServer's
 * P/Invoke declarations bind unchanged (LiveScanServer/KinectServer.cs:35-60, MainWindowForm.cs:42-43).
 * Part 2 is the device-resident API the same code runs on (plain pointers and sizes; no torch / HIP types):
 * what a host that already keeps frames in HBM (bench.py, the multi-GPU harness) calls.
 *
 * Nothing here throws; on failure the exports leave an empty mesh / untouched R,t and lsnGetLastError()
 * tells why.  There is NO CPU fallback: without a usable HIP device every compute entry point fails loudly.
 *
 * Paths in comments are relative to the reference repository root.
 */
#ifndef LSN_NATIVEUTILS_H
#define LSN_NATIVEUTILS_H

#include <stdbool.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---------------------------------------------------------------------------------------------------------
 * Part 1 -- the reference's exports
 * ------------------------------------------------------------------------------------------------------- */

/* include/NativeUtils/depthprocessing.h:29-33 -- 16 bytes, what KinectServer.CopyMeshToVerticesWithColoursArray
 * copies out (KinectServer.cs:376-389). */
typedef struct VertexC4ubV3f {
    unsigned char R, G, B, A;
    float X, Y, Z;
} VertexC4ubV3f;

/* include/NativeUtils/depthprocessing.h:42-48 (C# mirror LiveScanServer/Utils.cs:335-342) -- 32 bytes on LP64. */
typedef struct Mesh {
    int nVertices;
    VertexC4ubV3f *vertices;
    int nTriangles;
    int *triangles;
} Mesh;

/* include/NativeUtils/icp.h:15-18 */
typedef struct Point3f {
    float X, Y, Z;
} Point3f;

/* Replaces generateVerticesFromDepthMap, include/NativeUtils/depthprocessing.h:103-105
 * (src/NativeUtils/depthprocessing.cpp:1631-1657).  depth_maps / depth_colors are the concatenated per-sensor
 * buffers (u16 LE [h][w] / RGB8 [h][w][3]); intr_params 7 floats per sensor {cx,cy,fx,fy,r2,r4,r6};
 * wtransform_params 12 floats per sensor {t[3], R[3][3] row-major}, p' = R (p + t).
 * out_mesh->vertices is host memory owned by the library until deleteMesh; nTriangles = 0. */
void generateVerticesFromDepthMap(unsigned char *depth_maps, unsigned char *depth_colors, int *widths, int *heights,
                                  float *intr_params, float *wtransform_params, Mesh *out_mesh,
                                  float minX, float minY, float minZ, float maxX, float maxY, float maxZ,
                                  int depth_map_index);

/* Replaces generateMeshFromDepthMaps, include/NativeUtils/depthprocessing.h:108-110
 * (src/NativeUtils/depthprocessing.cpp:1715-1792).  In scope: the vertices of all sensors, cropped, in sensor
 * order then raster order, and the triangles of the reference's always-on triangulation (meshGenerator.cpp; indices
 * into `vertices`, reference order) -- i.e. the reference with both flags false.
 * bcolor_transfer / bgenerate_triangles select reference stages that are out of scope (colour transfer, overlay
 * merge); passing true is reported through lsnGetLastError() and otherwise ignored. */
void generateMeshFromDepthMaps(int n_maps, unsigned char *depth_maps, unsigned char *depth_colors, int *widths,
                               int *heights, float *intr_params, float *wtransform_params, Mesh *out_mesh,
                               bool bcolor_transfer, float minX, float minY, float minZ, float maxX, float maxY,
                               float maxZ, bool bgenerate_triangles);

/* Replaces depthMapAndColorSetRadialCorrection, include/NativeUtils/depthprocessing.h:111
 * (src/NativeUtils/depthprocessing.cpp:191-261,1794-1815): radial-distortion correction of every sensor's depth map and
 * colours, in place in the caller's host arrays (forward warp, last writer wins; raster-order in-place hole closing). */
void depthMapAndColorSetRadialCorrection(int n_maps, unsigned char *depth_maps, unsigned char *depth_colors, int *widths,
                                         int *heights, float *intr_params);

/* Extension for hosts that want one call per tick: LiveScanServer's tick is depthMapAndColorSetRadialCorrection followed by
 * generateMeshFromDepthMaps on the same arrays (LiveScanServer/KinectServer.cs:518-525, then :354-374), which sends the same
 * 8.7 MB of frames over PCIe twice.  This export does both with ONE upload: radial correction on the device, then the merge
 * call (flags false, false; vertices + triangles) on the corrected frames.  write_back_corrected != 0 also writes the corrected
 * maps back into depth_maps / depth_colors, as the separate export does (KinectServer keeps them, e.g. for recording);
 * 0 leaves the caller's arrays untouched.  Result = the two reference exports called one after the other, bit for bit. */
void lsnCorrectAndGenerateMesh(int n_maps, unsigned char *depth_maps, unsigned char *depth_colors, int *widths, int *heights,
                               float *intr_params, float *wtransform_params, Mesh *out_mesh, float minX, float minY, float minZ,
                               float maxX, float maxY, float maxZ, int write_back_corrected);

/* Host-only (no device needed), for tests and for operators who want to see it: the upload schedule a merge call (radial = 0), a
 * call that starts with the radial correction and goes on to the fusion (radial = 1: lsnCorrectAndGenerateMesh) or the radial export alone
 * (radial = 2: depthMapAndColorSetRadialCorrection) follows for sensors [first, first + count) of these frames, written to buf
 * as text -- e.g. "D[0-2] C[0-2] | D[3-7] C[3-5] | C[6-7]": runs of the depth / colour arrays in upload order (a pageable copy of
 * >= 1 MiB is pinned in place by the runtime, a smaller one staged at a quarter of the rate), `|` where a group of sensors is complete on
 * the device and launched.  sensors_per_group = 0: by size (the default); > 0: forced, like $LSN_HOST_GROUP.  Returns the number of
 * groups, -1 on bad arguments. */
int lsnHostScheduleDescribe(int n_maps, const int *widths, const int *heights, int first, int count, int radial, int sensors_per_group,
                            char *buf, int len);

/* Merge calls sharded over several devices.  $LSN_HOST_DEVICES=a,b,... (>= 2 entries, read at the first call; an entry may repeat):
 * generateMeshFromDepthMaps and lsnCorrectAndGenerateMesh give every listed device one contiguous block of the call's sensors -- uploaded
 * over that device's link, fused there, and stored by its kernels over its link into the SAME pinned Mesh blocks, behind the blocks
 * before it (formMesh's sensor order and triangle rebase, src/NativeUtils/depthprocessing.cpp:1594-1626; the reference's std::thread per
 * sensor, :708-733, with a device where it has a thread).  The mesh returned is byte for byte the one-device mesh.
 * lsnHostShardDescribe (host-only): how n_maps sensors are cut over n_devices (0 = as configured by the environment) -- text such as
 * "0:[0-3] 1:[4-7]" into buf, the block bounds into first_out (n_devices + 1 ints; may be NULL).  Returns the number of shards, -1 on
 * bad arguments. */
int lsnHostShardDescribe(int n_maps, int n_devices, int *first_out, char *buf, int len);
/* Measurement aid.  With $LSN_HOST_SHARD_SOLO=1 (read at the first sharded call) the parts of a sharded call run one after the other on
 * the calling thread, each alone on the link, and lsnHostShardPartMicros returns the wall time of every part of the last such call
 * (microseconds; `n` = room in `out`; returns the number of parts, 0 if there was no such call).  What a one-GPU box can say about a call
 * on several devices: each part's own cost on a link of its own; the call would take about as long as its slowest part. */
int lsnHostShardPartMicros(long long *out, int n);

/* Test hooks (no device needed).  lsnTestFaultPoints: how many fault points of a kind the process has passed (0 = guarded entries,
 * 1 = device / pinned allocations; $LSN_TEST_THROW / $LSN_TEST_FAIL_ALLOC = n make the n-th one throw std::bad_alloc inside the export).
 * lsnHostPoolStats: the pool of pinned mesh blocks -- blocks out with callers (a block a failed call did not return would stay here),
 * blocks waiting for reuse, bytes out; any pointer may be NULL. */
long long lsnTestFaultPoints(int kind);
int lsnHostPoolStats(int *live_blocks, int *pooled_blocks, long long *live_bytes);

/* Replaces createMesh / deleteMesh, src/NativeUtils/depthprocessing.cpp:1818-1835.  deleteMesh releases the two
 * arrays only (not the struct) and, unlike the reference, also nulls them so a second call is harmless. */
Mesh *createMesh(void);
void deleteMesh(Mesh *mesh);

/* Replaces ICP, include/NativeUtils/icp.h:65 (src/NativeUtils/icp.cpp:75-177): point-to-point ICP with exact
 * nearest neighbours, one-to-one matching, 2.5-sigma rejection on squared distances and an uncentred Kabsch step.
 * verts1 = target (read only), verts2 = source (moved in place), R (9, row-major) and t (3) in/out.
 * Returns 1.0f like the reference. */
float ICP(Point3f *verts1, Point3f *verts2, int nVerts1, int nVerts2, float *R, float *t, int maxIter);

/* ---------------------------------------------------------------------------------------------------------
 * Part 2 -- device-resident API (extension; every pointer named d_* is HIP device memory)
 * ------------------------------------------------------------------------------------------------------- */

/* Copies the calling thread's last error message (empty string = none) into buf; returns its length. */
int lsnGetLastError(char *buf, int len);

/* Number of visible HIP devices (0 when there is none or the runtime cannot initialise). */
int lsnDeviceCount(void);

/* Device memory and streams for hosts that speak nothing but this C-ABI (a C# or plain C host of the device-resident API has
 * no HIP of its own): thin wrappers over hipMalloc / hipFree / hipMemcpyAsync / hipStreamCreate / hipStreamSynchronize.
 * Copies are asynchronous on `stream` (0 = the null stream); host buffers must stay valid until the stream has been
 * synchronised.  All return 0 / a non-null pointer on success. */
void *lsnDeviceMalloc(int device, long long bytes);
int lsnDeviceFree(int device, void *d_ptr);
int lsnDeviceUpload(int device, void *d_dst, const void *h_src, long long bytes, void *stream);
int lsnDeviceDownload(int device, void *h_dst, const void *d_src, long long bytes, void *stream);
void *lsnStreamCreate(int device);
int lsnStreamDestroy(int device, void *stream);
int lsnStreamSynchronize(int device, void *stream);

/* A fusion plan: fixed rig geometry (n_maps sensors with widths/heights, as in the reference call) replicated
 * over n_ticks ticks that are fused by ONE launch sequence.  Holds the geometry tables and scan scratch in HBM. */
typedef struct LsnFusion LsnFusion;

LsnFusion *lsnFusionCreate(int device, int n_ticks, int n_maps, const int *widths, const int *heights);
void lsnFusionDestroy(LsnFusion *plan);

/* Vertices one tick can produce at most (= sum of w*h): the per-tick stride of d_vertices. */
long long lsnFusionTickCapacity(const LsnFusion *plan);

/* Host arrays exactly like the reference call: intr 7*n_maps, wt 12*n_maps, bounds {minX,minY,minZ,maxX,maxY,maxZ}.
 * stream is a hipStream_t passed as void* (NULL = the null stream). */
int lsnFusionSetParams(LsnFusion *plan, const float *intr_params, const float *wtransform_params,
                       const float *bounds6, void *stream);

/* Host-only (no device needed): the 16 floats the kernels read for one sensor -- {cx, cy, fx, fy, t[3], R[3][3] row-major} -- from the
 * caller's 7 intrinsics and 12 world-transform floats, exactly as lsnFusionSetParams packs them.  Exists so that the unpack order can be
 * held against the reference's IntrinsicCameraParameters(float*) / WorldTranformation(float*) constructors
 * (include/NativeUtils/depthprocessing.h:56-63,96-97) without a GPU. */
int lsnPackSensorParams(const float *intr7, const float *wt12, float *out16);

/* Selects how the raster-order compaction gets its global offsets: 0 = two-pass (count kernel + write kernel) -- except that a
 * ONE-tick plan of up to 2048 tiles takes the single pass of mode 2 by itself (one launch instead of three: 7.5-13.6 us per call
 * against 13.4-15.8; $LSN_ONE_TICK_SINGLE_PASS=0 / 1, read by lsnFusionCreate, forces either), 1 = single launch, runs of tiles
 * counted then re-evaluated, decoupled look-back per run, 2 = single pass, every tile evaluated once, decoupled look-back per
 * tile.  Results are identical; 0 is the fastest on MI355X.  The look-back forms (modes 1, 2 and the
 * one-tick case of mode 0) poll with a bound so that the grid always drains: a launch that gives up raises flag 1 of
 * lsnFusionCheck, writes nothing for the tiles concerned and -- single pass -- stores -1 as the tick's vertex count
 * (offsets[n_maps]) instead of leaving the previous call's value there. */
int lsnFusionSetMode(LsnFusion *plan, int mode);

/* Pipelined calls (mode 0 only): the caller promises that the INPUTS of a call are already resident and stay untouched
 * when the call is issued, whatever is still queued on its stream (true for double-buffered ingest).  The count + scan
 * of call k+1 then run on an internal side stream while the write kernel of call k is still running on the caller's
 * stream; outputs and their ordering on the caller's stream are unchanged. */
int lsnFusionSetPipelined(LsnFusion *plan, int enable);

/* Streamed calls: like lsnFusionRun (mode 0), but while this batch is written the NEXT batch's depth maps
 * (d_next_depth_maps, already resident; NULL = none) are counted by the same kernel -- the count pass is VALU-bound, the
 * write pass HBM-bound, so in one kernel they share every CU.  The following call with d_depth_maps == this call's
 * d_next_depth_maps skips its count pass.  Results are identical to lsnFusionRun. */
int lsnFusionRunStreamed(LsnFusion *plan, const void *d_depth_maps, const void *d_depth_colors, void *d_vertices, int *d_offsets,
                         const void *d_next_depth_maps, void *stream);

/* Fuses n_ticks ticks.  d_depth_maps: n_ticks x (concatenated u16 maps of one tick); d_depth_colors likewise RGB8;
 * d_vertices: n_ticks x lsnFusionTickCapacity() VertexC4ubV3f, tick k's merged cloud starts at k*capacity;
 * d_offsets: n_ticks x (n_maps+1) ints, [k][i] = index of sensor i's first vertex inside tick k's cloud,
 * [k][n_maps] = nVertices of tick k.  Asynchronous on `stream`; returns 0 on success. */
int lsnFusionRun(LsnFusion *plan, const void *d_depth_maps, const void *d_depth_colors, void *d_vertices,
                 int *d_offsets, void *stream);

/* The count pass needs no arithmetic once the calibration is fixed: from the second run with the same parameters on, the
 * plan keeps, per pixel, the interval of depth values that survive the transform + crop (found by bisection with the
 * pipeline's own roundings, exact by construction) and counts with two integer operations per pixel.  This call builds
 * the table now, reports the build time and optionally copies the table out (lsnFusionTickCapacity() words,
 * lo | count << 16; lo == 0: "evaluate arithmetically").  Returns 0, 1 when disabled ($LSN_NO_THRESHOLDS=1), -1 on error. */
int lsnFusionThresholds(LsnFusion *plan, unsigned int *out_host, float *build_ms, void *stream);

/* The complete merge call: vertices as lsnFusionRun plus the reference's always-on triangulation
 * (MeshGenerator::generateTrianglesGradients, src/NativeUtils/meshGenerator.cpp:14-181; index rebasing of formMesh,
 * src/NativeUtils/depthprocessing.cpp:1611-1627).  d_triangles: n_ticks x lsnFusionTickTriangleCapacity() x 3 ints
 * (vertex indices into the tick's merged cloud, reference order); d_tri_offsets: n_ticks x (n_maps+1) ints like d_offsets
 * ([k][n_maps] = nTriangles of tick k). */
long long lsnFusionTickTriangleCapacity(const LsnFusion *plan);
int lsnFusionRunMesh(LsnFusion *plan, const void *d_depth_maps, const void *d_depth_colors, void *d_vertices, int *d_offsets,
                     void *d_triangles, int *d_tri_offsets, void *stream);

/* Radial correction of n_ticks x n_maps frames in place in HBM (same layouts as lsnFusionRun's inputs);
 * intr_params: host, 7 floats per sensor {cx,cy,fx,fy,r2,r4,r6}. */
int lsnFusionRadialCorrect(LsnFusion *plan, const float *intr_params, void *d_depth_maps, void *d_depth_colors, void *stream);
/* The same out of place: the corrected frames go to d_depth_out / d_colors_out, the inputs stay as they are (the reference works
 * on copies and writes them back at the end, depthprocessing.cpp:193-194,259-260).  This is the cheaper form: the warped, not yet
 * closed maps then never leave the GPU's local memory (in place they pass through a scratch copy in HBM, because a pixel's source
 * may lie in rows another workgroup is already overwriting).  Both pointers equal to the inputs = lsnFusionRadialCorrect. */
int lsnFusionRadialCorrectTo(LsnFusion *plan, const float *intr_params, const void *d_depth_maps, const void *d_depth_colors,
                             void *d_depth_out, void *d_colors_out, void *stream);
/* Test hook: the number of work counters of the hole-closing chain that are NOT zero once `stream` has drained (0 after every complete
 * chain: the chain clears them itself, which is why the next call on the same stream needs no memset); -1 on error. */
int lsnFusionRadialCountersLeft(LsnFusion *plan, void *stream);

/* Name and average duration (ms, HIP events on the plan's stream) of the dominant kernel over the launches
 * since the last call with reset != 0; used by bench.py's roofline block.  Enable with lsnFusionProfile(plan, 1); lsnFusionProfile(plan, n)
 * with n > 1 times every n-th launch only (the two event records around a launch cost a few microseconds of stream time each). */
int lsnFusionProfile(LsnFusion *plan, int enable);
int lsnFusionKernelStats(LsnFusion *plan, double *avg_ms, long long *launches, char *name, int name_len, int reset);

/* Merged-cloud assembly after an all-gather of per-GPU shards (one block of sensors per GPU, rank order = sensor order):
 * d_shards [n_shards][n_ticks][shard_cap] vertices and d_shard_offsets [n_shards][n_ticks][maps_per_shard+1] are the
 * gathered outputs of lsnFusionRun; writes d_merged [n_ticks][merged_cap] (each tick contiguous, formMesh order,
 * src/NativeUtils/depthprocessing.cpp:1594-1608) and d_merged_offsets [n_ticks][n_shards*maps_per_shard+1]. */
int lsnMergeShards(int device, int n_shards, int n_ticks, int maps_per_shard, const void *d_shards, long long shard_cap,
                   const int *d_shard_offsets, void *d_merged, long long merged_cap, int *d_merged_offsets, void *stream);

/* Survivor exchange for the multi-GPU path: a vertex is 16 bytes, what it is computed from is 5 (u16 depth + RGB8) plus
 * one bit per pixel.  lsnFusionPackSurvivors fuses like lsnFusionRun but writes, per tick, the survivors' depth and colour
 * as compact streams in vertex order (d_depth_c [n_ticks][capacity] u16, d_rgb_c [n_ticks][capacity][3]), the survivor
 * mask (d_mask [n_ticks][capacity/8], bit p&7 of byte p>>3 = pixel p of the tick), the tiles' exclusive prefixes
 * (d_tile_prefix [n_ticks][lsnFusionTilesPerTick()]) and the usual offsets table.  After an all-gather of these arrays
 * lsnFusionReconstruct, called on a plan that covers the WHOLE rig (all sensors' parameters set, same n_ticks), rebuilds
 * every shard's vertices with the same arithmetic -- bit-identical to fusing all sensors in one plan -- directly into the
 * merged cloud d_merged [n_ticks][lsnFusionTickCapacity(all)] and fills d_merged_offsets [n_ticks][n_maps+1].  The
 * gathered arrays are [n_shards][n_ticks][...] with the per-tick vertex capacity of the streams cut to `slab`.
 * Needs identically sized sensors whose width is a multiple of 8; otherwise exchange vertices (lsnMergeShards). */
int lsnFusionTilesPerTick(const LsnFusion *plan);
int lsnFusionPackSurvivors(LsnFusion *plan, const void *d_depth_maps, const void *d_depth_colors, void *d_mask, void *d_depth_c,
                           void *d_rgb_c, int *d_tile_prefix, int *d_offsets, void *stream);
int lsnFusionReconstruct(LsnFusion *all, int n_shards, int maps_per_shard, const void *d_masks, const void *d_depth_c,
                         const void *d_rgb_c, long long slab, const int *d_tile_prefix, const int *d_shard_offsets, void *d_merged,
                         int *d_merged_offsets, void *stream);

/* The same two ends with the streams laid out the way lsnShardStep sends them: all ticks of a shard back to back, ONE
 * contiguous run per shard (d_depth_c / d_rgb_c as above but tick k's survivors start at d_tick_base[k], which the call
 * fills, [n_ticks] ints), so that a collective can send the run as it is.  The gathered streams are [n_shards][run_len]
 * (run_len >= the largest shard total); d_tick_base_scratch: [n_shards][n_ticks] ints. */
int lsnFusionPackSurvivorsRun(LsnFusion *plan, const void *d_depth_maps, const void *d_depth_colors, void *d_mask, void *d_depth_c,
                              void *d_rgb_c, int *d_tile_prefix, int *d_offsets, int *d_tick_base, void *stream);
int lsnFusionReconstructRun(LsnFusion *all, int n_shards, int maps_per_shard, const void *d_masks, const void *d_depth_c,
                            const void *d_rgb_c, long long run_len, const int *d_tile_prefix, const int *d_shard_offsets, void *d_merged,
                            int *d_merged_offsets, int *d_tick_base_scratch, void *stream);

/* ---------------------------------------------------------------------------------------------------------
 * Part 2b -- the multi-GPU step (one process per GPU): this rank's block of sensors in, the merged cloud of ALL sensors out.
 * Replaces the per-sensor thread fan-out and the concatenation of formMesh (src/NativeUtils/depthprocessing.cpp:708-733,
 * :1594-1608) across GPUs: rank r owns sensors [r * n_maps / world, (r + 1) * n_maps / world) of every tick (rank order =
 * sensor order); one exchange step over RCCL / xGMI (all-gathers of the survivors' inputs: 5 bytes per vertex + 1 bit per
 * pixel instead of 16 bytes per vertex) and every rank rebuilds the whole merged cloud, bit-identical to a single-GPU
 * lsnFusionRun over all sensors.  That exchange needs identically sized sensors whose width is a multiple of 8; any other rig
 * whose rank blocks hold the same number of pixels (or any rig with $LSN_SHARD_VERTICES=1) exchanges the 16-byte vertices
 * instead (lsnFusionRun on the block, one all-gather per tick cut to the step's largest shard, lsnMergeShards' packing pass):
 * the same merged cloud for about three times the bytes on the links.
 *
 * Rendezvous, in two steps so that no rank is left waiting inside a communicator that will never form: (1) every rank calls
 * lsnShardPrepare -- everything that can fail on one rank alone: argument checks, loading RCCL, device buffers -- and rank 0
 * calls lsnShardUniqueId; (2) the ranks agree, over whatever channel the host has (a file, a socket, torch.distributed), that ALL
 * of them are ready, and rank 0's 128 bytes go to everybody; (3) only then every rank calls lsnShardConnect with the same id
 * (collective: ncclCommInitRank blocks until all ranks have arrived).  lsnShardCreate = Prepare + Connect for hosts that have no
 * such channel.  widths / heights / intr / wt describe ALL n_maps sensors on every rank.
 * RCCL is loaded on first use: an instance already mapped in the process (PyTorch-ROCm's bundled one) is preferred, so a process
 * never holds two; lsnShardRcclPath reports the file.
 * lsnShardStep is collective and asynchronous on `stream` except for one event wait on a small pinned read-back (the element
 * count of a collective is a host-side argument; $LSN_SHARD_PADDED=1 trades it for full-capacity transfers).  All collectives
 * run on the caller's stream in one ncclGroup; $LSN_SHARD_CHUNKS=n (opt-in, not yet verified on a multi-GPU node) sends the
 * streams in n groups of ticks on a second stream beside the reconstruction.  After a failed step the handle refuses further
 * steps (the ranks' communicators are no longer in step): destroy it.
 * *d_merged: n_ticks x lsnShardMergedCapacity() VertexC4ubV3f, *d_merged_offsets: n_ticks x (n_maps + 1) ints, both owned
 * by the handle and valid until its next step. */
typedef struct LsnShard LsnShard;
int lsnShardUniqueId(unsigned char *id128);
LsnShard *lsnShardPrepare(int device, int rank, int world, int n_ticks, int n_maps, const int *widths, const int *heights);
int lsnShardConnect(LsnShard *shard, const unsigned char *id128);
LsnShard *lsnShardCreate(int device, int rank, int world, const unsigned char *id128, int n_ticks, int n_maps, const int *widths,
                         const int *heights);
int lsnShardRcclPath(char *buf, int len);
void lsnShardDestroy(LsnShard *shard);
long long lsnShardMergedCapacity(const LsnShard *shard);
int lsnShardSetParams(LsnShard *shard, const float *intr_all, const float *wt_all, const float *bounds6, void *stream);
int lsnShardStep(LsnShard *shard, const void *d_depth_local, const void *d_colors_local, void **d_merged, int **d_merged_offsets,
                 void *stream);
/* The handle's two plans, for lsnFusionProfile / lsnFusionKernelStats / lsnFusionCheck only (owned by the handle):
 * whole = 0: this rank's block of sensors (count / pack kernels), whole != 0: the whole rig (recon_kernel). */
LsnFusion *lsnShardPlan(LsnShard *shard, int whole);
/* bytes this rank contributed to the collectives of the last step (what every other rank received from it) */
long long lsnShardLastBytesSent(const LsnShard *shard);
/* the rank count the connected communicator itself reports (ncclCommCount; its ncclCommUserRank must equal the handle's rank);
 * -1 when the handle is not connected or the RCCL in use lacks the call */
int lsnShardRanksSeen(LsnShard *shard);

/* The plan's sticky device-side error flag since the last check (synchronises `stream`, clears the flag):
 *   0 = fine; 1 = a look-back launch (mode 1, mode 2, a one-tick plan of mode 0: lsnFusionSetMode) gave up on a bounded spin --
 *   the call's outputs are invalid (single pass: offsets[n_maps] = -1); 2 = a write pass found a tile whose survivors
 *   differ from what the count pass had counted (the inputs changed between the two passes: a buffer counted ahead by
 *   lsnFusionRunStreamed was refilled, or the inputs of a call in flight were overwritten) -- the affected tiles wrote
 *   nothing, the outputs of that call are invalid.  lsnFusionLookbackFailed is the same call under its round-1 name. */
int lsnFusionCheck(LsnFusion *plan, void *stream);
int lsnFusionLookbackFailed(LsnFusion *plan, void *stream);

/* An ICP workspace for clouds of at most max_n1 target / max_n2 source points. */
typedef struct LsnIcp LsnIcp;

LsnIcp *lsnIcpCreate(int device, int max_n1, int max_n2);
void lsnIcpDestroy(LsnIcp *icp);

/* nn_mode: 0 = brute force (one query per lane, the targets streamed through scalar loads; the ablation leg),
 *          1 = voxel grid (exact at any distance: near path per query + box hierarchy per group of 64 queries; when a work
 *              list overflows, the complete hierarchy walk per group).  Both give the same bits.
 * d_verts1: n1*3 floats (target), d_verts2: n2*3 floats (source, moved in place), d_R 9 floats, d_t 3 floats
 * (in/out, device).  Asynchronous on `stream`; no host synchronisation inside the iteration loop. */
int lsnIcpRun(LsnIcp *icp, const float *d_verts1, int n1, float *d_verts2, int n2, float *d_R, float *d_t,
              int maxIter, int nn_mode, void *stream);

/* The NN step alone (parity tests, ablation): d_idx n2 ints, d_dist2 n2 floats. */
int lsnIcpNearest(LsnIcp *icp, const float *d_verts1, int n1, const float *d_verts2, int n2, int *d_idx,
                  float *d_dist2, int nn_mode, void *stream);

/* The whole pose-refinement pass of LiveScanServer (refineWorker_DoWork, LiveScanServer/MainWindowForm.cs:330-410) in one
 * call: Gauss-Seidel over the sensors x n_refine_iters, each step ICP(all other sensors' current clouds, this sensor's
 * cloud, Rs[i], Ts[i], n_icp_iters), with every cloud resident in HBM for the whole pass.  clouds[i] = counts[i] x 3 floats
 * on the HOST, moved in place; world_R (n x 9) / world_t (n x 3), nullable, are updated like worldTransforms[i]
 * (:382-410, the C# loops as written); Rs_out (n x 9) / Ts_out (n x 3), nullable, receive the accumulated ICP poses.
 * The pass's device state (ICP workspace, cloud buffers, stream) is kept for the next call on the same device (grown when a
 * rig needs more); a pass that runs while another is in flight allocates its own and frees it again. */
int lsnRefine(int device, int n_sensors, float *const *clouds, const int *counts, int n_refine_iters, int n_icp_iters,
              float *world_R, float *world_t, float *Rs_out, float *Ts_out);

/* How many queries of the workspace's last voxel-grid NN step were settled by the near path (the walk over the cells around
 * a query whose bound is small; diagnostic, synchronises `stream`; -1 on error).  $LSN_ICP_NEAR (read by lsnIcpCreate):
 * 0 = the path off, every query goes through the group search; 1 (default) = the unseeded step always probes, a seeded step takes
 * the path when at least half of the previous step's queries lay within one target cell of their neighbour; 2 = always.
 * $LSN_ICP_NEAR_PTS: the candidate cap per query (<= 128).  Speed only: all settings give the same bits. */
int lsnIcpNearResolved(LsnIcp *icp, void *stream);

/* Optional phase timing of lsnIcpRun (measurement aid): with profiling on, every lsnIcpRun records HIP events on its stream;
 * lsnIcpProfile synchronises `stream` and returns the milliseconds of the last run in ms4 = {grid build + source sort,
 * NN steps (incl. the fused apply of the previous iteration), match statistics + Kabsch sums + solve, final apply}. */
int lsnIcpSetProfiling(LsnIcp *icp, int on);
int lsnIcpProfile(LsnIcp *icp, float *ms4, void *stream);

/* Per-iteration diagnostics of the last lsnIcpRun (copied to host; synchronises `stream`):
 * out[iter] = {n_matched, n_kept, mean, stddev, T[3], Rn[9]} as 16 floats (counts stored as floats). */
int lsnIcpTrace(LsnIcp *icp, float *out16_per_iter, int max_iters, void *stream);

/* The reference's tick, device resident, as ONE call: the radial correction (out of place) then the merge call with its
 * triangulation -- CorrectRadialDistortionsForDepthMaps then GenerateMesh on every tick (LiveScanServer/KinectServer.cs:518-525,
 * :354-374) = lsnFusionRadialCorrectTo + lsnFusionRunMesh on a batch of n_ticks ticks.  From 8 ticks up the batch is cut in two
 * halves that run side by side -- the caller's stream and an internal one, joined before the call's work ends on the caller's
 * stream, the second half started behind the first half's band kernel -- so that one half's latency chains (the closing rounds)
 * are filled by the other half's throughput-bound passes: +2 % ticks/s on scene frames with the join per call (+5 % for two
 * free-running streams, which a caller that can consume the halves separately may build from two plans itself);
 * $LSN_TICK_PARTS=1: one plan, one stream.  Same bytes as the two calls on one plan.  Arrays as for those calls: inputs and corrected maps [n_ticks][pixels per tick] u16 / [..][3] u8,
 * d_vertices [n_ticks][lsnTickCapacity()] VertexC4ubV3f, d_triangles [n_ticks][lsnTickTriangleCapacity()][3] int,
 * offset tables [n_ticks][n_maps + 1]. */
typedef struct LsnTick LsnTick;
LsnTick *lsnTickCreate(int device, int n_ticks, int n_maps, const int *widths, const int *heights);
void lsnTickDestroy(LsnTick *tick);
int lsnTickSetParams(LsnTick *tick, const float *intr_params, const float *wtransform_params, const float *bounds6, void *stream);
long long lsnTickCapacity(const LsnTick *tick);
long long lsnTickTriangleCapacity(const LsnTick *tick);
int lsnTickParts(const LsnTick *tick);   /* 1 or 2 */
int lsnTickRun(LsnTick *tick, const void *d_depth_in, const void *d_colors_in, void *d_depth_corrected, void *d_colors_corrected,
               void *d_vertices, int *d_offsets, void *d_triangles, int *d_tri_offsets, void *stream);

/* ------------------------------------------------------------------------------------------------------------------
 * Part 3 -- the data formats either side of the path.
 *
 * Outbound, built on the device from a cloud / mesh that is already in HBM (e.g. the outputs of lsnFusionRunMesh):
 * the byte stream TransferSocket.SendFrame writes (LiveScanServer/TransferSocket.cs:50-104) for the chunks
 * TransferServer forms (formMeshChunks, LiveScanServer/TransferServer.cs:203-270, when there are triangles, else
 * formVerticesChunks, :177-201), and the file Utils.saveToPly(binary) writes (LiveScanServer/Utils.cs:222-262).
 * ------------------------------------------------------------------------------------------------------------------ */
typedef struct LsnTransfer LsnTransfer;

/* Workspace for clouds of at most max_vertices vertices / max_triangles triangles.  Like LsnFusion and LsnIcp a handle
 * serialises its own calls (internal mutex); independent handles run concurrently. */
LsnTransfer *lsnTransferCreate(int device, int max_vertices, int max_triangles);
void lsnTransferDestroy(LsnTransfer *t);

/* Upper bound on the length of the SendFrame stream. */
long long lsnTransferFrameBound(int n_vertices, int n_triangles);

/* d_vertices: n_vertices VertexC4ubV3f, d_triangles: n_triangles x 3 ints (may be null when n_triangles == 0), both on
 * the device.  Writes to d_out (device, 4-byte aligned, out_cap bytes):
 *   int nVertices, int nTriangles, int nChunks, int vChunkSizes[nChunks], int tChunkSizes[nChunks],
 *   float xyz[3*nVertices], u8 rgb[3*nVertices], int tri[3*nTriangles]
 * where, as in formMeshChunks, vertices are re-emitted per chunk in order of first use, triangle indices are
 * chunk-local, a chunk closes at the first triangle end with >= 64997 vertices (and the first chunk's triangle count
 * is one short when there are several chunks, TransferServer.cs:244).  Returns the stream length in bytes, -1 on
 * error (an index outside [0, n_vertices) is an error; the reference would throw).  Synchronises `stream`. */
long long lsnTransferPack(LsnTransfer *t, const void *d_vertices, int n_vertices, const int *d_triangles, int n_triangles,
                          void *d_out, long long out_cap, void *stream);

/* How the last lsnTransferPack of this handle formed its chunks (a measurement / test aid; the bytes do not depend on it):
 * 0 = vertices only (formVerticesChunks), 1 = all chunks at once from one prefix sum over the index positions (meshes whose
 * vertices are each used at most 16 times, consecutive uses < 64997 index positions apart: every grid mesh), 2 = chunk after
 * chunk (any mesh; what 1 falls back to).  -1 on a null handle. */
int lsnTransferLastPath(LsnTransfer *t);

/* Binary PLY: header + 15-byte vertex records {f32 x,y,z; u8 r,g,b} + 13-byte face records {u8 3; i32 a,b,c}.
 * lsnPlyBinaryBytes gives the exact file length; lsnPlyPack writes the file image to d_out (device, any alignment),
 * asynchronously on `stream`, and returns its length (-1 on error). */
long long lsnPlyBinaryBytes(int n_vertices, int n_triangles);
long long lsnPlyPack(int device, const void *d_vertices, int n_vertices, const int *d_triangles, int n_triangles, void *d_out,
                     long long out_cap, void *stream);

/* Host callers (LiveScanServer): the mesh the last generateMeshFromDepthMaps / generateVerticesFromDepthMap /
 * lsnCorrectAndGenerateMesh call OF THE CALLING THREAD returned is still in HBM -- or its frames are, and it is rebuilt there on demand
 * (a call sharded over $LSN_HOST_DEVICES: its frames are gathered from the devices first); these build its SendFrame stream / binary
 * PLY image on the device and copy the bytes to `out` (host).  With out == NULL they return an upper bound on (stream) / the exact (PLY)
 * length.  Return the length, -1 on error.
 * Which mesh: the one of the calling thread's own last mesh call (merge calls and single-sensor calls run on different workers in
 * LiveScanServer, MainWindowForm.cs:238,304: the refine thread's single-sensor call does not change what the update thread gets);
 * a thread that has made no mesh call gets the process's last one.  A LATER call of the same family from another thread still
 * replaces the mesh -- one thread per family, as in LiveScanServer. */
long long lsnLastMeshTransferFrame(unsigned char *out, long long out_cap);
long long lsnLastMeshPly(unsigned char *out, long long out_cap);

/* Inbound (host-side parsing, no device work): the client's frame message -- LiveScanClient::SerializeFrame
 * (src/LiveScanClient/liveScanClient.cpp:185-290) as KinectSocket.ReceiveFrame reads it
 * (LiveScanServer/KinectSocket.cs:211-304): 16-byte header {i32 payload bytes, i32 compressed, i32 w, i32 h}, then
 * the payload (one zstd frame when compressed == 1) = u16 depth[w*h], u8 rgb[3*w*h], body block.  zstd is the system
 * libzstd.so.1, loaded on first use (the reference P/Invokes libzstd.dll, LiveScanServer/ZSTDDecompressor.cs:13-31). */
typedef struct LsnFrameInfo { int payload_bytes, compressed, width, height; } LsnFrameInfo;

int lsnZstdAvailable(void);
/* 0 = a frame follows, 1 = "no more frames" (payload_bytes <= 0, KinectSocket.cs:231-235), -1 = malformed. */
int lsnFrameParseHeader(const unsigned char *header16, LsnFrameInfo *info);
/* Copies depth (w*h*2 bytes), rgb (w*h*3) and the body block (at most bodies_cap bytes) out of a payload; any of the
 * three outputs may be null.  Returns the length of the body block (>= 4), -1 on error. */
long long lsnFrameDecode(const unsigned char *payload, int payload_bytes, int compressed, int width, int height,
                         unsigned char *depth_out, unsigned char *rgb_out, unsigned char *bodies_out, int bodies_cap,
                         int *n_bodies);
/* The inverse (SerializeFrame after its colour mapping): header + payload into out; compression_level 0 = raw,
 * > 0 = zstd at that level.  bodies may be null (= no bodies).  Returns the message length, -1 on error. */
long long lsnFrameEncode(const unsigned char *depth, const unsigned char *rgb, int width, int height, const unsigned char *bodies,
                         int bodies_bytes, int compression_level, unsigned char *out, long long out_cap);

/* The client's recording file (src/LiveScanClient/frameFileWriterReader.cpp:59-82 reader, :115-130 writer): records
 * "bufferSize= %d\nframe_timestamp= %d\n" + frame message + "\n", parsed from / appended to a memory image of the
 * file.  lsnRecordingNext returns the position after the record at `pos` (-1 at the end of the file or on a malformed
 * record -- then lsnGetLastError is non-empty); lsnRecordingAppend returns the bytes written (-1 when out is too small). */
long long lsnRecordingNext(const unsigned char *file, long long len, long long pos, long long *frame_off, int *frame_len,
                           int *timestamp_ms);
long long lsnRecordingAppend(unsigned char *out, long long cap, const unsigned char *frame, int len, int timestamp_ms);

#ifdef __cplusplus
}
#endif
#endif
