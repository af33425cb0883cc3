// mesh.hip -- the always-on triangulation of the merge call: tri_kernel<0/1> + lsnFusionRunMesh.
// Shares the plan and the tile machinery of fusion.hip (fusion_shared.hpp).
#include "fusion_shared.hpp"

namespace {

// ---- triangulation (the "next" row after the vertex path) --------------------------------------------------------
//
// Replaces MeshGenerator::generateTrianglesGradients (src/NativeUtils/meshGenerator.cpp:14-181, driver
// depthprocessing.cpp:1659-1691) and formMesh's triangle part (:1611-1627).  Per pixel with a vertex, a 2x2 stencil
// (P, U = up, UR = up-right, R = right) yields up to two triangles after depth-continuity tests that also look one
// step further along every edge; the reference's 4 row-band threads concatenated in order are plain raster order over
// y in [2, h-2), x in [1, w-2).  Same structure as the vertex path: count -> scan_kernel -> write, 8 pixels per lane,
// the 4 x 11 depth window and the 2 x 9 index window of a lane live in registers.  All integer arithmetic.

// (TriArgs, the kernel argument block, is declared in fusion_shared.hpp next to FuseArgs)

constexpr int kTriWinDefault = 1536;  // triangles staged per LDS round (18 KB); a multiple of 16 (LSN_TRI_WINDOW: 256 .. 4096)
// The staged window is padded by one int per 16 triangles: a lane of 8 pixels on a closed surface holds 16 triangles, so the k-th
// triangles of the lanes of a wave lie 48 ints apart -- 4 banks for 64 lanes (SQ_LDS_BANK_CONFLICT: 77 % of the LDS cycles of the
// write pass, which made it LDS-bound); 49 ints apart they take 64 different banks.
__host__ __device__ constexpr int stage_ints(int win) { return 3 * win + win / 16 + 8; }
__device__ __forceinline__ int stage_slot(int lead, int r) { return lead + 3 * r + (r >> 4); }                 // where triangle r of the window starts
__device__ __forceinline__ int stage_phys(int lead, int i) { return i + (int)((unsigned int)(i - lead) / 48u); }  // where int i >= lead of the unpadded window lies

// MeshGenerator::checkTriangleConstraints (meshGenerator.cpp:14-61) on a pixel's 4 x 4 depth window W[dy + 2][dx + 1],
// dx in [-1, 2], dy in [-2, 1]; the three corners are compile-time offsets, so are the forward / backward probes.
// depth_thr = (int)((v0 + v1 + v2) / 3.0 * 0.00272 + 7.273) (:26, double) equals (272 s + 2181900) / 300000 in integers
// for EVERY possible sum s of three u16 -- proven exhaustively in tests/test_fast_division.py.
template <int X1, int Y1, int X2, int Y2>
__device__ __forceinline__ bool edge_ok(const int (&W)[4][4], int thr)
{
    const int val1 = W[Y1 + 2][X1 + 1], val2 = W[Y2 + 2][X2 + 1];
    if (abs(val1 - val2) < thr) return true;                                  // :35-36
    constexpr int SX = X2 - X1, SY = Y2 - Y1;
    const int val_forward = W[Y2 + SY + 2][X2 + SX + 1];                      // :39-40
    if (val_forward != 0 && abs(val2 - val1 - (val_forward - val2)) < thr) return true;    // :42-47
    const int val_backward = W[Y1 - SY + 2][X1 - SX + 1];                     // :50
    if (val_backward != 0 && abs(val2 - val1 - (val1 - val_backward)) < thr) return true;  // :51-56
    return false;
}

template <int X1, int Y1, int X2, int Y2, int X3, int Y3>
__device__ __forceinline__ bool tri_ok(const int (&W)[4][4])
{
    const int v0 = W[Y1 + 2][X1 + 1], v1 = W[Y2 + 2][X2 + 1], v2 = W[Y3 + 2][X3 + 1];
    if (v0 == 0 || v1 == 0 || v2 == 0) return false;                          // :22-23
    const int thr = (272 * (v0 + v1 + v2) + 2181900) / 300000;
    return edge_ok<X1, Y1, X2, Y2>(W, thr) && edge_ok<X2, Y2, X3, Y3>(W, thr) && edge_ok<X3, Y3, X1, Y1>(W, thr);
}

// Which of the four candidate triangles of a pixel are emitted: bit i = triangle i of meshGenerator.cpp:101-104
// (0: R,U,P  1: R,UR,U  2: P,UR,U  3: P,R,UR), after the vertex-index checks of :133-134.
__device__ __forceinline__ unsigned int pixel_triangles(const int (&W)[4][4], int mP, int mU, int mUR, int mR)
{
    if (mP == -1) return 0;                                                   // :113-114
    const bool t0 = tri_ok<0, 0, 0, -1, 1, 0>(W);                             // :117
    const bool t1 = tri_ok<1, 0, 0, -1, 1, -1>(W);                            // :118
    bool t2 = false, t3 = false;
    if (!t0 && !t1) {
        t2 = tri_ok<0, 0, 0, -1, 1, -1>(W);                                   // :122
        t3 = tri_ok<0, 0, 1, -1, 1, 0>(W);                                    // :123
    }
    unsigned int m = 0;
    if (t0 && mR != -1 && mU != -1) m |= 1u;
    if (t1 && mR != -1 && mUR != -1 && mU != -1) m |= 2u;
    if (t2 && mUR != -1 && mU != -1) m |= 4u;
    if (t3 && mR != -1 && mUR != -1) m |= 8u;
    return m;
}

// The same verdicts for a lane's 8 consecutive pixels of one row, with every edge evaluated once.  checkTriangleConstraints
// accepts an edge when ANY of three differences is below the triangle's threshold (:35-56), and the three differences do not
// depend on the direction the edge is walked in (walking B->A swaps the roles of the forward and the backward probe), so an
// undirected edge has ONE metric = their minimum and passes for a triangle iff metric < that triangle's threshold.  The four
// candidate triangles of a pixel share 5 edges (and the vertical one with the next pixel): 5 metrics per pixel instead
// of 12 edge walks, no branches.  D: depth rows y-2 .. y+1, columns x0-1 .. x0+9; M: vertex indices of rows y-1, y.
// depth_thr as tri_ok computes it: (272 s + 2181900) / 300000 (:26, in integers)

// "metric < tri_threshold(s)" without the division: metric + 1 <= floor((272 s + 2181900) / 300000)  <=>  300000 (metric + 1) <= 272 s + 2181900
// <=>  18750 metric <= 17 s + 117618 (divide by 16, floor the constant: both sides are integers).  metric <= 65535 and s <= 3 * 65535, so both
// sides fit 32 bits (18750 * 65535 < 2^31) and their factors 24 -- v_mul_u32_u24 / v_mad_u32_u24, full rate, where the division by a constant
// was a quarter-rate v_mul_hi_u32 plus shifts per threshold, four thresholds per pixel.  A triangle passes when ALL THREE of its edge metrics
// are below its threshold, i.e. when their maximum is: one v_max3_u32, one multiply, one compare per triangle instead of three compares.
// Equivalence with tri_threshold for every s and every metric: tests/test_fast_division.py.
constexpr unsigned int kThrMul = 18750u, kThrSum = 17u, kThrAdd = 117618u;
// edge_metric with the probes pre-biased: Z = depth + 2^17 for a valid probe pixel, 2^30 for an invalid one (depth 0), so
// that |x - probe| becomes one v_sad_u32 on non-negative operands and an invalid probe yields a difference no threshold
// can reach -- no select per edge.  (2 vB - vA) + 2^17 and (2 vA - vB) + 2^17 lie in [65537, 262142].
constexpr unsigned int kProbeBias = 1u << 17, kProbeInvalid = 1u << 30;

__device__ __forceinline__ unsigned int abs_diff_u32(unsigned int a, unsigned int b)
{
    unsigned int r;
    asm("v_sad_u32 %0, %1, %2, 0" : "=v"(r) : "v"(a), "v"(b));   // |a - b| in one VALU slot (the compiler has no pattern for it here)
    return r;
}

// The depth window of a lane as it is loaded: per row the pixel left of the lane's eight (0 = there is none), the eight as four packed
// pairs, the two to their right as one packed pair.
struct DepthRow {
    unsigned int left;
    unsigned int mid[4];
    unsigned int right;
};

// depth | bias of window column c (0 .. 10) of a row, straight from the packed pairs: ONE instruction per value (v_and_or_b32 for the low
// half of a pair, v_perm_b32 for the high half: bytes 2,3 of the pair under bytes 2,3 of the bias) where unpacking and biasing were two.
// Nothing below needs the raw depth: differences of two biased values are the differences of the depths, "depth != 0" is "value != bias",
// and the three-depth sum of a triangle carries 3 x bias, which the threshold's constant takes back (kThrAddBiased).
template <int C>
__device__ __forceinline__ unsigned int biased_depth(const DepthRow &r)
{
    if (C == 0) return r.left | kProbeBias;
    const unsigned int pair = C <= 8 ? r.mid[(C - 1) >> 1] : r.right;
    const bool high = C <= 8 ? ((C - 1) & 1) != 0 : C == 10;
    return high ? __builtin_amdgcn_perm(kProbeBias, pair, 0x07060302u) : ((pair & 0xFFFFu) | kProbeBias);
}

// The edge metric on biased values: zA, zB the edge's ends (depth | bias), pBeyondB / pBeyondA the probes one step further (depth | bias, or
// kProbeInvalid for a depth of 0 -- a difference no threshold reaches).  With d = zB - zA: 2 vB - vA + bias = zB + d, 2 vA - vB + bias = zA - d.
__device__ __forceinline__ unsigned int edge_metric_z(unsigned int zA, unsigned int zB, unsigned int pBeyondB, unsigned int pBeyondA)
{
    const unsigned int d = zB - zA;
    const unsigned int a = abs_diff_u32(zA, zB);                    // |vB - vA|                 (:35)
    const unsigned int f = abs_diff_u32(zB + d, pBeyondB);          // |d - (beyondB - vB)|      (:39-47)
    const unsigned int b = abs_diff_u32(zA - d, pBeyondA);          // |d - (vA - beyondA)|      (:50-56)
    return min(a, min(f, b));
}

// edges_pass for a sum of three BIASED depths: 17 (s - 3 bias) + 117618 = 17 s + (117618 - 51 bias); the constant is negative, the sum is
// not (s >= 3 bias), so the wrapped 32-bit arithmetic gives the true value.
constexpr unsigned int kThrAddBiased = kThrAdd - 51u * kProbeBias;
__device__ __forceinline__ bool edges_pass_z(unsigned int m0, unsigned int m1, unsigned int m2, unsigned int s_biased)
{
    return __umul24(max(m0, max(m1, m2)), kThrMul) <= __umul24(s_biased, kThrSum) + kThrAddBiased;
}

// a + b + c as one v_add3_u32 (left alone the compiler shares a two-term sum between two triangles and spends six adds on four sums).
// (Tried and dropped: collecting a verdict with one v_addc_co_u32 -- mask + mask + carry-in -- instead of a select and an or: the verdict
// has to become a lane mask operand first, which costs the compiler a select and a compare: 972 against 941 VALU instructions per lane.)
__device__ __forceinline__ unsigned int add3(unsigned int a, unsigned int b, unsigned int c)
{
    unsigned int r;
    asm("v_add3_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

// bit k of an 8-bit mask -> bit 4 k
__device__ __forceinline__ unsigned int spread_to_nibbles(unsigned int x)
{
    x = (x | (x << 12)) & 0x000F000Fu;
    x = (x | (x << 6)) & 0x03030303u;
    return (x | (x << 3)) & 0x11111111u;
}

// mU9 / mP9: bit c = the pixel of column x0 + c in row y - 1 / row y has a vertex (c = 0 .. 8).  rows[0 .. 3] = depth rows y-2 .. y+1.
// The per-pixel verdicts t0 .. t3 of the four candidate triangles are collected as four 8-bit masks (bit k = pixel k); everything that
// does not depend on the depths -- which pixels have vertices (:113-114, :133-134), which columns count (:87-90), "2 and 3 only when
// neither 0 nor 1 passed" (:120) -- is then a handful of bit operations per LANE instead of per pixel, and the masks are interleaved into
// the 4-bits-per-pixel code word the write pass reads.
__device__ __forceinline__ unsigned int lane_triangles(const DepthRow (&rows)[4], unsigned int mU9, unsigned int mP9, int x0, int w)
{
    unsigned int Z[4][kPxPerLane + 3], Pz[4][kPxPerLane + 3];   // depth | bias; the same as a probe (invalid -> kProbeInvalid)
    bool ok[4][kPxPerLane + 3];
#define LSN_ROW(R)                                                                                         \
    {                                                                                                      \
        Z[R][0] = biased_depth<0>(rows[R]); Z[R][1] = biased_depth<1>(rows[R]); Z[R][2] = biased_depth<2>(rows[R]);    \
        Z[R][3] = biased_depth<3>(rows[R]); Z[R][4] = biased_depth<4>(rows[R]); Z[R][5] = biased_depth<5>(rows[R]);    \
        Z[R][6] = biased_depth<6>(rows[R]); Z[R][7] = biased_depth<7>(rows[R]); Z[R][8] = biased_depth<8>(rows[R]);    \
        Z[R][9] = biased_depth<9>(rows[R]); Z[R][10] = biased_depth<10>(rows[R]);                                      \
    }
    LSN_ROW(0) LSN_ROW(1) LSN_ROW(2) LSN_ROW(3)
#undef LSN_ROW
#pragma unroll
    for (int r = 0; r < 4; r++)
#pragma unroll
        for (int c = 0; c < kPxPerLane + 3; c++) {
            ok[r][c] = Z[r][c] != kProbeBias;
            Pz[r][c] = ok[r][c] ? Z[r][c] : kProbeInvalid;
        }
    unsigned int ev[kPxPerLane + 1];   // P-U of window column c = 1 .. 9
#pragma unroll
    for (int c = 1; c <= kPxPerLane + 1; c++) ev[c - 1] = edge_metric_z(Z[2][c], Z[1][c], Pz[0][c], Pz[3][c]);
    unsigned int T0 = 0, T1 = 0, T2 = 0, T3 = 0;
#pragma unroll
    for (int k = 0; k < kPxPerLane; k++) {
        const int c = k + 1;
        const unsigned int zP = Z[2][c], zU = Z[1][c], zUR = Z[1][c + 1], zR = Z[2][c + 1];
        const unsigned int hP = edge_metric_z(zP, zR, Pz[2][c + 2], Pz[2][c - 1]);    // P - R
        const unsigned int hU = edge_metric_z(zU, zUR, Pz[1][c + 2], Pz[1][c - 1]);   // U - UR
        const unsigned int d1 = edge_metric_z(zU, zR, Pz[3][c + 2], Pz[0][c - 1]);    // U - R   (down-right)
        const unsigned int d2 = edge_metric_z(zP, zUR, Pz[0][c + 2], Pz[3][c - 1]);   // P - UR  (up-right)
        const unsigned int pu = ev[c - 1], ru = ev[c];                                 // P - U, R - UR
        const bool vP = ok[2][c], vU = ok[1][c], vUR = ok[1][c + 1], vR = ok[2][c + 1];   // :22-23
        const bool t0 = vR & vU & vP & edges_pass_z(d1, pu, hP, add3(zP, zR, zU));     // R,U,P   (:117)
        const bool t1 = vR & vUR & vU & edges_pass_z(ru, hU, d1, add3(zU, zUR, zR));   // R,UR,U  (:118)
        const bool t2 = vP & vUR & vU & edges_pass_z(d2, hU, pu, add3(zU, zUR, zP));   // P,UR,U  (:122; "only when neither 0 nor 1": below)
        const bool t3 = vP & vR & vUR & edges_pass_z(hP, ru, d2, add3(zP, zR, zUR));   // P,R,UR  (:123)
        T0 |= t0 ? (1u << k) : 0u;
        T1 |= t1 ? (1u << k) : 0u;
        T2 |= t2 ? (1u << k) : 0u;
        T3 |= t3 ? (1u << k) : 0u;
    }
    // which of the lane's columns count (:87-90: 1 <= x < w - 2), which pixels have a vertex (:113-114)
    const int lo = x0 < 1 ? 1 - x0 : 0, hi = min(kPxPerLane, w - 2 - x0);
    const unsigned int cols = hi > lo ? ((1u << hi) - 1u) & ~((1u << lo) - 1u) : 0u;
    const unsigned int P8 = mP9 & cols & 0xFFu, R8 = (mP9 >> 1) & 0xFFu, U8 = mU9 & 0xFFu, UR8 = (mU9 >> 1) & 0xFFu;
    const unsigned int alt = ~(T0 | T1);                                                // :120, on the verdicts themselves (before :133-134)
    const unsigned int V0 = T0 & P8 & R8 & U8;                                          // :133-134
    const unsigned int V1 = T1 & P8 & R8 & UR8 & U8;
    const unsigned int V2 = T2 & alt & P8 & UR8 & U8;
    const unsigned int V3 = T3 & alt & P8 & R8 & UR8;
    return spread_to_nibbles(V0) | (spread_to_nibbles(V1) << 1) | (spread_to_nibbles(V2) << 2) | (spread_to_nibbles(V3) << 3);
}

#ifndef LSN_TRI_MIN_WAVES
#define LSN_TRI_MIN_WAVES 5   // waves per SIMD the count pass is compiled for (build-time, A/B: 5 = 102 VGPRs allowed, 7 = 73; it needs 71)
#endif
// MODE 0 = count triangles per tile, 1 = write them at the scanned offsets.
// HOST (write pass): `tri` is pinned host memory -- plain stores, rounds aligned to the destination (stage_and_store's note).  A template
// parameter, not a run-time flag: a run-time choice between a streaming and a plain store of the same value to the same address is
// folded into ONE plain store by the compiler, and the device-resident path loses its streaming stores (65.8 against 71.7 k ticks/s).
template <int MODE, bool VEC, bool HOST = false>
__global__ __launch_bounds__(kThreads, MODE == 0 ? LSN_TRI_MIN_WAVES : 1) void tri_kernel(const TriArgs a)
{
    extern __shared__ int stage[];   // write pass: stage_ints(a.win) + 3 * 64 ints
    const int kTriWin = a.win;
    __shared__ int s_wave_tot[4];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int tick = blockIdx.x / a.tiles_per_tick;
    const int tile = blockIdx.x - tick * a.tiles_per_tick;
    const TileDesc td = a.tiles[tile];
    const FrameDesc fd = a.frames[td.frame];
    const int w = fd.w, h = fd.h;
    const unsigned short *dep = a.depth + tick * a.tick_pix_stride + fd.depth_off;
    const int *map = a.pixmap + tick * a.tick_pix_stride + fd.depth_off;
    const int p0 = (tile - fd.tile_start) * kTile + threadIdx.x * kPxPerLane;
    const bool in_frame = p0 < fd.npix;

    // (x, y) of the lane's first pixel, as in compute_tile
    const int v = td.x0 + (int)threadIdx.x * kPxPerLane;
    int q = (int)((float)v * fd.inv_w);
    int x0 = v - q * w;
    if (x0 < 0) { q--; x0 += w; }
    if (x0 >= w) { q++; x0 -= w; }
    int y0 = td.y0 + q;
    if (!in_frame) { x0 = 0; y0 = 0; }

    unsigned int code = 0;      // 4 bits per pixel: which triangles it emits
    int M[2][kPxPerLane + 1];   // vertex indices: row y-1 (U, UR) and row y (P, R), columns x0 .. x0+8
#pragma unroll
    for (int r = 0; r < 2; r++)
#pragma unroll
        for (int c = 0; c <= kPxPerLane; c++) M[r][c] = -1;
    // VEC: the compact map of rows y-1 and y -- the lane's own group of 8 pixels and the first pixel of the group to its right
    int first[2] = {0, 0}, first_next[2] = {0, 0};
    unsigned int mask9[2] = {0, 0};
    auto load_groups = [&]() {
#pragma unroll
        for (int r = 0; r < 2; r++) {
            const long long g = (tick * a.tick_pix_stride + fd.depth_off + (long long)(y0 - 1 + r) * w + x0) >> 3;
            first[r] = a.pm_first[g];
            unsigned int m = a.pm_mask[g];
            if (x0 + 8 < w) {
                first_next[r] = a.pm_first[g + 1];
                m |= ((unsigned int)a.pm_mask[g + 1] & 1u) << 8;
            }
            mask9[r] = m;
        }
    };

    const size_t code_slot = (size_t)blockIdx.x * kThreads + threadIdx.x;
    if (MODE == 1) {
        // the count pass already evaluated every stencil: reload its verdicts, rebuild only the vertex indices
        code = a.codes[code_slot];
        if (VEC && code != 0) {
            load_groups();
#pragma unroll
            for (int r = 0; r < 2; r++) {
                // a pixel without a vertex gets the index of the next one: never read, its triangles are not in `code` (:133-134)
                int run = first[r] + a.index_base;
#pragma unroll
                for (int c = 0; c < kPxPerLane; c++) {
                    M[r][c] = run;
                    run += (int)((mask9[r] >> c) & 1u);
                }
                M[r][8] = first_next[r] + a.index_base;
            }
        }
    } else if (VEC) {
        // w % 8 == 0: the 8 pixels share row y0; window rows y0-2 .. y0+1, columns x0-1 .. x0+9
        const bool row_ok = in_frame && y0 >= 2 && y0 < h - 2;   // :87-90 (bands clamp to [2, h-2))
        bool any_vertex = false;   // a pixel without a vertex emits nothing (:113-114): a lane whose 8 pixels have none skips the stencils --
        if (row_ok) {               // on real frames whole rows outside the crop box do, i.e. whole waves
            load_groups();
            any_vertex = (mask9[1] & 0xFFu) != 0;
        }
        if (any_vertex) {
            DepthRow rows[4];
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const unsigned short *row = dep + (long long)(y0 - 2 + r) * w;
                const uint4 c = *reinterpret_cast<const uint4 *>(row + x0);
                rows[r].left = x0 > 0 ? row[x0 - 1] : 0;
                rows[r].mid[0] = c.x; rows[r].mid[1] = c.y; rows[r].mid[2] = c.z; rows[r].mid[3] = c.w;
                rows[r].right = x0 + 8 < w ? *reinterpret_cast<const unsigned int *>(row + x0 + 8) : 0u;
            }
            code = lane_triangles(rows, mask9[0], mask9[1], x0, w);
        }
    } else {
        // general widths: a lane's pixels may span rows; every pixel fetches its own 4 x 4 window
        int x = x0, y = y0;
#pragma unroll
        for (int k = 0; k < kPxPerLane; k++) {
            if (p0 + k < fd.npix && y >= 2 && y < h - 2 && x >= 1 && x < w - 2) {
                int W[4][4];
#pragma unroll
                for (int r = 0; r < 4; r++)
#pragma unroll
                    for (int c = 0; c < 4; c++) W[r][c] = dep[(long long)(y - 2 + r) * w + (x - 1 + c)];
                const long long p = (long long)y * w + x;
                M[1][k] = map[p];            // P
                M[0][k] = map[p - w];        // U
                // UR / R of this pixel are kept in the slots the VEC path would use only when they do not collide:
                // the general path re-reads them at emission time instead (see below), the code word is what counts
                code |= pixel_triangles(W, map[p], map[p - w], map[p - w + 1], map[p + 1]) << (4 * k);
            }
            x++;
            if (x == w) { x = 0; y++; }
        }
    }

    // ---- ranks ---------------------------------------------------------------------------------------------------
    const int cnt = __popc(code);
    const int incl = wave_inclusive_scan(cnt, lane);
    if (lane == 63) s_wave_tot[wave] = incl;
    int base = 0;
    if (MODE == 1) base = a.tile_counts[blockIdx.x];
    __syncthreads();
    int wave_off = 0, tile_tot = 0;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int t = s_wave_tot[i];
        if (i < wave) wave_off += t;
        tile_tot += t;
    }
    if (MODE == 0) {
        a.codes[code_slot] = code;
        if (threadIdx.x == 0) a.tile_counts[blockIdx.x] = tile_tot;
        return;
    }

    // ---- stage in rank order, copy out coalesced (triangles_shifts order, meshGenerator.cpp:101-104) ------------------
    int *dst = a.tri + 3 * (tick * a.tick_tri_stride + base);
    // the window is staged shifted by the destination's misalignment (0..3 ints; the same for every window: a window is a
    // multiple of 16 bytes), so that it leaves as whole 16-byte chunks -- 4-byte stores before: scene frames write 1.2 GB of
    // triangles per 64 ticks
    const int lead = (int)((reinterpret_cast<uintptr_t>(dst) >> 2) & 3);
    const int rank0 = wave_off + incl - cnt;
    for (int w0 = 0; w0 < tile_tot; w0 += kTriWin) {
        int r = rank0 - w0;
        int x = x0, y = y0;
        // a pixel emits triangles {0}, {1}, {0, 1}, {2}, {3} or {2, 3} of :101-104 (2 and 3 only when neither 0 nor 1 passed, :120):
        // at most two, a first and a second one, each three selects -- not four predicated slots.  Lanes whose triangles lie
        // outside this window sit the round out (ranks ascend with the lane: whole waves do).
        if (cnt != 0 && r < kTriWin && r + cnt > 0) {
            if (VEC) {
                // no branches: a triangle that does not exist, or lies outside the window, goes to a slot of the lane's own behind it
                const int nowhere = stage_ints(kTriWin) + 3 * lane;
#pragma unroll
                for (int k = 0; k < kPxPerLane; k++) {
                    const unsigned int m = (code >> (4 * k)) & 15u;
                    const int mP = M[1][k], mU = M[0][k], mUR = M[0][k + 1], mR = M[1][k + 1];
                    const bool b0 = (m & 1u) != 0, b12 = (m & 6u) != 0, low = (m & 3u) != 0, two = (m == 3u) | (m == 12u);
                    const int o1 = ((m != 0) & ((unsigned int)r < (unsigned int)kTriWin)) ? stage_slot(lead, r) : nowhere;
                    stage[o1] = low ? mR : mP;                                         // 0: R,U,P  1: R,UR,U  2: P,UR,U  3: P,R,UR
                    stage[o1 + 1] = b0 ? mU : (b12 ? mUR : mR);
                    stage[o1 + 2] = b0 ? mP : (b12 ? mU : mUR);
                    r += m != 0;
                    const int o2 = (two & ((unsigned int)r < (unsigned int)kTriWin)) ? stage_slot(lead, r) : nowhere;
                    stage[o2] = low ? mR : mP;                                         // the second one: 1 after 0, 3 after 2
                    stage[o2 + 1] = low ? mUR : mR;
                    stage[o2 + 2] = low ? mU : mUR;
                    r += two;
                }
            } else {
#pragma unroll
                for (int k = 0; k < kPxPerLane; k++) {
                    const unsigned int m = (code >> (4 * k)) & 15u;
                    if (m) {
                        const long long p = (long long)y * w + x;
                        // (every index a code bit names exists, :133-134: no -1 is rebased)
                        const int mP = map[p] + a.index_base, mU = map[p - w] + a.index_base, mUR = map[p - w + 1] + a.index_base, mR = map[p + 1] + a.index_base;
                        const bool b0 = (m & 1u) != 0, b12 = (m & 6u) != 0, low = (m & 3u) != 0;
                        if ((unsigned int)r < (unsigned int)kTriWin) {
                            const int o = stage_slot(lead, r);
                            stage[o] = low ? mR : mP;
                            stage[o + 1] = b0 ? mU : (b12 ? mUR : mR);
                            stage[o + 2] = b0 ? mP : (b12 ? mU : mUR);
                        }
                        r++;
                        if (m == 3u || m == 12u) {
                            if ((unsigned int)r < (unsigned int)kTriWin) {
                                const int o = stage_slot(lead, r);
                                stage[o] = low ? mR : mP;
                                stage[o + 1] = low ? mUR : mR;
                                stage[o + 2] = low ? mU : mUR;
                            }
                            r++;
                        }
                    }
                    x++;
                    if (x == w) { x = 0; y++; }
                }
            }
        }
        __syncthreads();
        const int n = 3 * min(kTriWin, tile_tot - w0);   // ints, at stage[lead ..)
        int *out = dst + 3 * w0;
        {
            const int end = lead + n;                      // in ints, relative to the aligned start of the first chunk
            const int c0 = lead ? 1 : 0, c1 = end >> 2;     // chunks [c0, c1) are whole
            int4 *g16 = reinterpret_cast<int4 *>(out - lead);
            const int mis = HOST ? (int)((reinterpret_cast<uintptr_t>(g16 + c0) >> 4) & 63) : 0;
            for (int j = c0 + (int)threadIdx.x - mis; j < c1; j += kThreads) {
                if (HOST && j < c0) continue;
                int4 v;                                     // written once, never read again by this launch sequence
                const unsigned int o = (unsigned int)(4 * j - lead), q = o / 48u, rem = o - 48u * q;   // one division per chunk: a pad may fall inside it
                const int ph = 4 * j + (int)q;
                v.x = stage[ph];
                v.y = stage[ph + 1 + (rem + 1 >= 48u)];
                v.z = stage[ph + 2 + (rem + 2 >= 48u)];
                v.w = stage[ph + 3 + (rem + 3 >= 48u)];
                if (HOST) {
                    g16[j] = v;
                } else {
                    __builtin_nontemporal_store(v.x, &g16[j].x);
                    __builtin_nontemporal_store(v.y, &g16[j].y);
                    __builtin_nontemporal_store(v.z, &g16[j].z);
                    __builtin_nontemporal_store(v.w, &g16[j].w);
                }
            }
            const int head = lead ? min(n, 4 - lead) : 0;   // the ragged ends: the neighbouring tiles own the rest of those chunks
            const int tail0 = max(head, 4 * c1 - lead);
            if ((int)threadIdx.x < head) out[threadIdx.x] = stage[stage_phys(lead, lead + threadIdx.x)];
            if (tail0 + (int)threadIdx.x < n) out[tail0 + threadIdx.x] = stage[stage_phys(lead, lead + tail0 + threadIdx.x)];
        }
        __syncthreads();
    }
}


}  // namespace

// The triangle passes of a tick whose pixel -> vertex map is filled, in two halves (count -> scan | write); p->mu held.
static bool tri_args(LsnFusion *p, const void *d_depth, void *d_triangles, TriArgs &t)
{
    t.frames = p->frames.as<FrameDesc>();
    t.tiles = p->tile_frame.as<TileDesc>();
    t.depth = static_cast<const unsigned short *>(d_depth);
    t.pixmap = p->pixmap.as<int>();
    t.pm_first = p->pm_first.as<int>();
    t.pm_mask = p->pm_mask.as<unsigned char>();
    t.tri = static_cast<int *>(d_triangles);
    t.tile_counts = p->tri_counts.as<int>();
    t.codes = p->tri_codes.as<unsigned int>();
    t.tiles_per_tick = p->tiles_per_tick;
    static const int win_env = getenv("LSN_TRI_WINDOW") ? atoi(getenv("LSN_TRI_WINDOW")) : kTriWinDefault;
    t.win = std::min(4096, std::max(256, win_env)) & ~15;
    t.host_out = 0;
    t.index_base = 0;
    t.tick_pix_stride = p->cap;
    t.tick_tri_stride = 2 * p->cap;
    return p->pixmap_compact && ((uintptr_t)d_depth & 15) == 0;   // the vertex pass wrote the compact map iff it ran its wide-load form
}

static int triangle_count_passes(LsnFusion *p, const void *d_depth, int *d_tri_offsets, hipStream_t s, const lsn::RunHooks *hooks)
{
    TriArgs t;
    const bool vec = tri_args(p, d_depth, nullptr, t);
    const int grid = p->tiles_per_tick * p->n_ticks;
    if (vec) hipLaunchKernelGGL((tri_kernel<0, true>), dim3(grid), dim3(kThreads), 0, s, t);
    else     hipLaunchKernelGGL((tri_kernel<0, false>), dim3(grid), dim3(kThreads), 0, s, t);
    const bool mirror = hooks && hooks->mirror && hooks->h_tri_offsets;
    hipLaunchKernelGGL(scan_kernel, dim3((unsigned)p->n_ticks), dim3(kScanThreads), 0, s, t.tile_counts, t.tiles_per_tick, t.frames, p->n_maps,
                       d_tri_offsets, mirror ? hooks->h_tri_offsets : nullptr);
    if (hooks && hooks->h_tri_offsets && !mirror)
        LSN_HIP(hipMemcpyAsync(hooks->h_tri_offsets, d_tri_offsets, sizeof(int) * (size_t)p->n_ticks * (p->n_maps + 1), hipMemcpyDeviceToHost, s));
    if (hooks && hooks->tri_counted) LSN_HIP(hipEventRecord(hooks->tri_counted, s));
    LSN_HIP(hipGetLastError());
    return 0;
}

static int triangle_write_pass(LsnFusion *p, const void *d_depth, void *d_triangles, int index_base, bool host_out, hipStream_t s)
{
    TriArgs t;
    const bool vec = tri_args(p, d_depth, d_triangles, t);
    t.host_out = host_out;
    t.index_base = index_base;
    const size_t stage_bytes = sizeof(int) * (size_t)(stage_ints(t.win) + 3 * 64);
    const int grid = p->tiles_per_tick * p->n_ticks;
    if (host_out) {
        if (vec) hipLaunchKernelGGL((tri_kernel<1, true, true>), dim3(grid), dim3(kThreads), stage_bytes, s, t);
        else     hipLaunchKernelGGL((tri_kernel<1, false, true>), dim3(grid), dim3(kThreads), stage_bytes, s, t);
    } else {
        if (vec) hipLaunchKernelGGL((tri_kernel<1, true>), dim3(grid), dim3(kThreads), stage_bytes, s, t);
        else     hipLaunchKernelGGL((tri_kernel<1, false>), dim3(grid), dim3(kThreads), stage_bytes, s, t);
    }
    LSN_HIP(hipGetLastError());
    return 0;
}

static int triangle_passes(LsnFusion *p, const void *d_depth, void *d_triangles, int *d_tri_offsets, hipStream_t s, const lsn::RunHooks *hooks)
{
    if (triangle_count_passes(p, d_depth, d_tri_offsets, s, hooks)) return -1;
    return triangle_write_pass(p, d_depth, d_triangles, 0, hooks && hooks->host_out, s);
}

extern "C" long long lsnFusionTickTriangleCapacity(const LsnFusion *p) { return p ? 2 * p->cap : 0; }

static int lsnFusionRunMesh_impl(LsnFusion *p, const void *d_depth, const void *d_colors, void *d_vertices, int *d_offsets,
                                void *d_triangles, int *d_tri_offsets, void *stream)
{
    lsn::clear_error();
    return lsn::run_mesh(p, d_depth, d_colors, d_vertices, d_offsets, d_triangles, d_tri_offsets, lsn::as_stream(stream), nullptr);
}

extern "C" int lsnFusionRunMesh(LsnFusion *p, const void *d_depth, const void *d_colors, void *d_vertices, int *d_offsets,
                                void *d_triangles, int *d_tri_offsets, void *stream)
{
    return lsn::guarded<int>("lsnFusionRunMesh", static_cast<int>(-1), [&]() { return lsnFusionRunMesh_impl(p, d_depth, d_colors, d_vertices, d_offsets, d_triangles, d_tri_offsets, stream); });
}

int lsn::run_mesh(LsnFusion *p, const void *d_depth, const void *d_colors, void *d_vertices, int *d_offsets, void *d_triangles,
                  int *d_tri_offsets, hipStream_t s, const RunHooks *hooks)
{
    if (!p || !d_depth || !d_colors || !d_vertices || !d_offsets || !d_triangles || !d_tri_offsets) {
        lsn::set_error("lsnFusionRunMesh: null argument");
        return -1;
    }
    // one critical section for the vertex pass (which fills the pixel -> vertex map) and the triangle passes (which read it):
    // two threads sharing a plan cannot interleave between them
    std::lock_guard<std::mutex> g(p->mu);
    LSN_HIP(hipSetDevice(p->device));
    // the pixel -> vertex map: 5 bytes per 8 pixels when the vertex pass runs its wide-load form (the same test as in run_locked),
    // else one int per pixel
    const bool wide = p->vec_ok && ((uintptr_t)d_depth & 15) == 0 && ((uintptr_t)d_colors & 7) == 0 && (p->tick_depth_elems % 8) == 0;
    if ((wide ? (p->pm_first.reserve(sizeof(int) * ((size_t)p->cap * p->n_ticks / 8 + 2)) || p->pm_mask.reserve((size_t)p->cap * p->n_ticks / 8 + 2))
              : p->pixmap.reserve(sizeof(int) * (size_t)p->cap * p->n_ticks)) ||
        p->tri_counts.reserve(sizeof(int) * (size_t)p->tiles_per_tick * p->n_ticks) ||
        p->tri_codes.reserve(sizeof(unsigned int) * (size_t)p->tiles_per_tick * p->n_ticks * kThreads))
        return -1;
    // vertices + depth_to_vertices_map (count / scan / write launches)
    if (lsn::run_locked(p, d_depth, d_colors, d_vertices, d_offsets, s, true, hooks)) return -1;
    return triangle_passes(p, d_depth, d_triangles, d_tri_offsets, s, hooks);
}

int lsn::run_triangles(LsnFusion *p, const void *d_depth, void *d_triangles, int *d_tri_offsets, int *tri_mirror, bool host_out, hipStream_t s,
                       hipEvent_t tri_counted)
{
    if (!p || !d_depth || !d_triangles || !d_tri_offsets) {
        lsn::set_error("run_triangles: null argument");
        return -1;
    }
    std::lock_guard<std::mutex> g(p->mu);
    LSN_HIP(hipSetDevice(p->device));
    if (p->tri_counts.reserve(sizeof(int) * (size_t)p->tiles_per_tick * p->n_ticks) ||
        p->tri_codes.reserve(sizeof(unsigned int) * (size_t)p->tiles_per_tick * p->n_ticks * kThreads))
        return -1;
    lsn::RunHooks hooks;
    hooks.mirror = tri_mirror != nullptr;
    hooks.host_out = host_out;
    hooks.h_tri_offsets = tri_mirror;
    hooks.tri_counted = tri_counted;
    return triangle_passes(p, d_depth, d_triangles, d_tri_offsets, s, &hooks);
}


// The same in two calls, for a caller that has to know the tick's triangle count before it can say where the triangles go (host_flows.hip: the
// sensor blocks of a call sharded over devices).  Nothing else may run on the plan in between (the lane's lock).
int lsn::run_triangles_count(LsnFusion *p, const void *d_depth, int *d_tri_offsets, int *tri_mirror, hipEvent_t tri_counted, hipStream_t s)
{
    if (!p || !d_depth || !d_tri_offsets) {
        lsn::set_error("run_triangles_count: null argument");
        return -1;
    }
    std::lock_guard<std::mutex> g(p->mu);
    LSN_HIP(hipSetDevice(p->device));
    if (p->tri_counts.reserve(sizeof(int) * (size_t)p->tiles_per_tick * p->n_ticks) ||
        p->tri_codes.reserve(sizeof(unsigned int) * (size_t)p->tiles_per_tick * p->n_ticks * kThreads))
        return -1;
    lsn::RunHooks hooks;
    hooks.mirror = tri_mirror != nullptr;
    hooks.h_tri_offsets = tri_mirror;
    hooks.tri_counted = tri_counted;
    return triangle_count_passes(p, d_depth, d_tri_offsets, s, &hooks);
}

int lsn::run_triangles_write(LsnFusion *p, const void *d_depth, void *d_triangles, int index_base, bool host_out, hipStream_t s)
{
    if (!p || !d_depth || !d_triangles) {
        lsn::set_error("run_triangles_write: null argument");
        return -1;
    }
    std::lock_guard<std::mutex> g(p->mu);
    LSN_HIP(hipSetDevice(p->device));
    return triangle_write_pass(p, d_depth, d_triangles, index_base, host_out, s);
}
