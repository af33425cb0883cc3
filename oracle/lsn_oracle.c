/*
 * lsn_oracle.c -- CPU restatement (plain C) of LiveScan3D NativeUtils' fusion hot path.
 *
 * TEST INFRASTRUCTURE ONLY -- see lsn_oracle.h for the rules and the pinning status.
 * Build: gcc -O2 -ffp-contract=off -fopenmp (oracle/Makefile).  No -march=native / -mfma:
 * the reference is built /fp:precise (NativeUtils/NativeUtils.vcxproj:283-301), every f32
 * operation is rounded on its own.
 */
#include "lsn_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------------------------------
 * depth -> cloud
 * ---------------------------------------------------------------------------------------- */

/* createVertices, src/NativeUtils/depthprocessing.cpp:122-187.  Operation order, line by line:
 *   :144      skip depth 0
 *   :149-150  Z = float(d) / 1000.0f
 *   :151-152  X = (x - cx) / fx ; Y = (cy - y) / fy        (int -> float, then f32 sub, f32 div)
 *   :154-155  X *= Z ; Y *= Z
 *   :157-159  += t
 *   :160      RotatePoint (:109-120): res_i = (X*Ri0 + Y*Ri1) + Z*Ri2
 *   :162      inclusive AABB, NaN passes
 *   :166-175  raster-order compaction
 * and formMesh's repack :1598-1604 (A = 255). */
int orc_create_vertices(const uint16_t *depth, const uint8_t *rgb, int w, int h,
                        const float *intr7, const float *wt12, const float *bounds6,
                        orc_vertex *out, int *vert_to_pix, int *pix_to_vert)
{
    const float cx = intr7[0], cy = intr7[1], fx = intr7[2], fy = intr7[3];
    const float t0 = wt12[0], t1 = wt12[1], t2 = wt12[2];
    const float *R = wt12 + 3;
    const float minX = bounds6[0], minY = bounds6[1], minZ = bounds6[2];
    const float maxX = bounds6[3], maxY = bounds6[4], maxZ = bounds6[5];
    int n = 0;

    if (pix_to_vert)
        for (long i = 0; i < (long)w * h; i++) pix_to_vert[i] = -1;

    for (int y = 0; y < h; y++) {
        const uint16_t *row = depth + (long)y * w;
        for (int x = 0; x < w; x++) {
            if (row[x] == 0) continue;
            long pos = x + (long)y * w;
            float val = (float)row[x];
            float Z = val / 1000.0f;
            float X = ((float)x - cx) / fx;
            float Y = (cy - (float)y) / fy;
            X = X * Z;
            Y = Y * Z;
            X = X + t0;
            Y = Y + t1;
            Z = Z + t2;
            float rx = X * R[0] + Y * R[1] + Z * R[2];
            float ry = X * R[3] + Y * R[4] + Z * R[5];
            float rz = X * R[6] + Y * R[7] + Z * R[8];
            if (rx < minX || rx > maxX || ry < minY || ry > maxY || rz < minZ || rz > maxZ)
                continue;
            if (pix_to_vert) pix_to_vert[pos] = n;
            if (vert_to_pix) vert_to_pix[n] = (int)pos;
            out[n].R = rgb[pos * 3];
            out[n].G = rgb[pos * 3 + 1];
            out[n].B = rgb[pos * 3 + 2];
            out[n].A = 255;
            out[n].X = rx;
            out[n].Y = ry;
            out[n].Z = rz;
            n++;
        }
    }
    return n;
}

long orc_generate_mesh_vertices(int n_maps, const uint8_t *depth_maps, const uint8_t *depth_colors,
                                const int *widths, const int *heights,
                                const float *intr, const float *wt, const float *bounds6,
                                orc_vertex *out, int *per_map_counts, int n_threads)
{
    /* generateVerticesFromDepthMaps :708-733 -- buffer striding += w*h*2 / += w*h*3 */
    long *dpos = (long *)malloc(sizeof(long) * (size_t)(n_maps + 1) * 3);
    long *cpos = dpos + (n_maps + 1), *vpos = cpos + (n_maps + 1);
    int *cnt = (int *)malloc(sizeof(int) * (size_t)(n_maps > 0 ? n_maps : 1));
    dpos[0] = cpos[0] = vpos[0] = 0;
    for (int i = 0; i < n_maps; i++) {
        long np = (long)widths[i] * heights[i];
        dpos[i + 1] = dpos[i] + np * 2;
        cpos[i + 1] = cpos[i] + np * 3;
        vpos[i + 1] = vpos[i] + np;
    }
    /* per-sensor scratch at its worst-case slot, compacted afterwards in sensor order (formMesh :1594-1608) */
    orc_vertex *scratch = (orc_vertex *)malloc(sizeof(orc_vertex) * (size_t)(vpos[n_maps] > 0 ? vpos[n_maps] : 1));
    (void)n_threads;
#pragma omp parallel for schedule(static, 1) num_threads(n_threads > 0 ? n_threads : 1)
    for (int i = 0; i < n_maps; i++) {
        cnt[i] = orc_create_vertices((const uint16_t *)(depth_maps + dpos[i]), depth_colors + cpos[i],
                                     widths[i], heights[i], intr + 7 * i, wt + 12 * i, bounds6,
                                     scratch + vpos[i], NULL, NULL);
    }
    long total = 0;
    for (int i = 0; i < n_maps; i++) {
        memcpy(out + total, scratch + vpos[i], sizeof(orc_vertex) * (size_t)cnt[i]);
        total += cnt[i];
        if (per_map_counts) per_map_counts[i] = cnt[i];
    }
    free(scratch);
    free(cnt);
    free(dpos);
    return total;
}

int orc_generate_vertices_from_depth_map(const uint8_t *depth_maps, const uint8_t *depth_colors,
                                         const int *widths, const int *heights,
                                         const float *intr, const float *wt, const float *bounds6,
                                         int depth_map_index, orc_vertex *out)
{
    /* :1646-1650 */
    long depth_pos = 0, colors_pos = 0;
    for (int i = 0; i < depth_map_index; i++) {
        depth_pos += (long)widths[i] * heights[i] * 2;
        colors_pos += (long)widths[i] * heights[i] * 3;
    }
    return orc_create_vertices((const uint16_t *)(depth_maps + depth_pos), depth_colors + colors_pos,
                               widths[depth_map_index], heights[depth_map_index],
                               intr + 7 * depth_map_index, wt + 12 * depth_map_index, bounds6,
                               out, NULL, NULL);
}

/* ------------------------------------------------------------------------------------------
 * radial correction
 * ---------------------------------------------------------------------------------------- */

/* (int) of a float the way the reference's x86-64 build does it (cvttss2si): truncation toward zero,
 * NaN and values outside int range give INT_MIN ("integer indefinite"). */
static int f2i_x86(float v)
{
    if (!(v > -2147483904.0f && v < 2147483648.0f)) return (int)0x80000000;
    return (int)v;
}

void orc_radial_correction(uint16_t *depth_map, uint8_t *colors, int w, int h, const float *intr7)
{
    const float cx = intr7[0], cy = intr7[1], fx = intr7[2], fy = intr7[3];
    const float r2 = intr7[4], r4 = intr7[5], r6 = intr7[6];                    /* :196-198 */
    const long np = (long)w * h;
    uint16_t *map_copy = (uint16_t *)calloc((size_t)(np > 0 ? np : 1), sizeof(uint16_t));
    uint8_t *colors_copy = (uint8_t *)calloc((size_t)(np > 0 ? np : 1) * 3, 1);

    for (int y = 0; y < h; y++)                                                  /* :200-218 forward warp */
        for (int x = 0; x < w; x++) {
            if (depth_map[x + (long)y * w] == 0) continue;
            float u = ((float)x - cx) / fx;
            float v = ((float)y - cy) / fy;
            float r = u * u + v * v;
            float d = 1 - r2 * r - r4 * r * r - r6 * r * r * r;
            int x_corr = f2i_x86(u * d * fx + cx);
            int y_corr = f2i_x86(v * d * fy + cy);
            if (x_corr >= 0 && y_corr >= 0 && x_corr < w && y_corr < h) {
                map_copy[x_corr + (long)y_corr * w] = depth_map[x + (long)y * w];
                memcpy(colors_copy + (x_corr + (long)y_corr * w) * 3, colors + (x + (long)y * w) * 3, 3);
            }
        }

    /* :223-256 closing holes, in place and in raster order */
    const int shifts[8] = {-w - 1, -w, -w + 1, -1, 1, w - 1, w, w + 1};
    for (int y = 1; y < h - 1; y++)
        for (int x = 1; x < w - 1; x++) {
            long pos = x + (long)y * w;
            int val = map_copy[pos];
            if (val == 0) {
                int n = 0, sum = 0, sum_cR = 0, sum_cG = 0, sum_cB = 0, prev_val = -1;
                for (int i = 0; i < 8; i++)
                    if ((map_copy[pos + shifts[i]] > 0) && (prev_val == -1 || abs(map_copy[pos + shifts[i]] - prev_val) < 30)) {
                        prev_val = map_copy[pos + shifts[i]];
                        n++;
                        sum += map_copy[pos + shifts[i]];
                        sum_cR += colors_copy[(pos + shifts[i]) * 3];
                        sum_cG += colors_copy[(pos + shifts[i]) * 3 + 1];
                        sum_cB += colors_copy[(pos + shifts[i]) * 3 + 2];
                    }
                if (n > 4) {
                    map_copy[pos] = (uint16_t)(sum / n);
                    colors_copy[pos * 3] = (uint8_t)(sum_cR / n);
                    colors_copy[pos * 3 + 1] = (uint8_t)(sum_cG / n);
                    colors_copy[pos * 3 + 2] = (uint8_t)(sum_cB / n);
                }
            }
        }
    memcpy(depth_map, map_copy, (size_t)np * sizeof(uint16_t));                  /* :259-260 */
    memcpy(colors, colors_copy, (size_t)np * 3);
    free(colors_copy);
    free(map_copy);
}

void orc_radial_correction_all(int n_maps, uint8_t *depth_maps, uint8_t *depth_colors, const int *widths, const int *heights,
                               const float *intr, int n_threads)
{
    long *dpos = (long *)malloc(sizeof(long) * (size_t)(n_maps + 1) * 2);
    long *cpos = dpos + (n_maps + 1);
    dpos[0] = cpos[0] = 0;
    for (int i = 0; i < n_maps; i++) {
        dpos[i + 1] = dpos[i] + (long)widths[i] * heights[i] * 2;
        cpos[i + 1] = cpos[i] + (long)widths[i] * heights[i] * 3;
    }
    (void)n_threads;
#pragma omp parallel for schedule(static, 1) num_threads(n_threads > 0 ? n_threads : 1)
    for (int i = 0; i < n_maps; i++)                                              /* :1803-1814, one thread per map */
        orc_radial_correction((uint16_t *)(depth_maps + dpos[i]), depth_colors + cpos[i], widths[i], heights[i], intr + 7 * i);
    free(dpos);
}

/* ------------------------------------------------------------------------------------------
 * triangulation
 * ---------------------------------------------------------------------------------------- */

/* MeshGenerator::checkTriangleConstraints, src/NativeUtils/meshGenerator.cpp:14-61.
 * p1,p2,p3 are linear pixel positions inside `depth`. */
static int tri_ok(const uint16_t *depth, long p1, long p2, long p3)
{
    const int vals[3] = {depth[p1], depth[p2], depth[p3]};
    const long ptrs[3] = {p1, p2, p3};
    static const int pairs_1[3] = {0, 1, 2};
    static const int pairs_2[3] = {1, 2, 0};
    if (vals[0] == 0 || vals[1] == 0 || vals[2] == 0) return 0;                         /* :22-23 */
    /* :26  (int)((v0+v1+v2) / 3.0 * 0.00272 + 7.273), evaluated in double */
    const int depth_thr = (int)((vals[0] + vals[1] + vals[2]) / 3.0 * 0.00272 + 7.273);
    for (int tr = 0; tr < 3; tr++) {
        const int ind1 = pairs_1[tr], ind2 = pairs_2[tr];
        const int val1 = vals[ind1], val2 = vals[ind2];
        if (abs(val1 - val2) < depth_thr) continue;                                      /* :35-36 */
        const long shift = ptrs[ind2] - ptrs[ind1];                                      /* :39 */
        const int val_forward = depth[ptrs[ind2] + shift];                               /* :40 */
        if (val_forward != 0) {
            const int gradient_forward = val_forward - val2;
            if (abs(val2 - val1 - gradient_forward) < depth_thr) continue;               /* :44-46 */
        }
        const int val_backward = depth[ptrs[ind1] - shift];                              /* :50 */
        if (val_backward != 0) {
            const int gradient_backward = val1 - val_backward;
            if (abs(val2 - val1 - gradient_backward) < depth_thr) continue;              /* :53-55 */
        }
        return 0;                                                                        /* :57 */
    }
    return 1;
}

long orc_generate_triangles(const uint16_t *depth, const int *pix_to_vert, int w, int h, int index_base, int *out)
{
    /* generateTrianglesGradientsRegion, meshGenerator.cpp:77-144, with the 4 row bands of :147-181 run back to back:
     * every band clamps to y in [max(.,2), min(.,h-2)) and x in [1, w-2), bands tile [0,h) in order. */
    long n = 0;
    const int sh_up = -w, sh_upright = -w + 1, sh_right = 1;                              /* :93 pixel_shifts */
    const int tshift[12] = {sh_right, sh_up, 0,   sh_right, sh_upright, sh_up,           /* :101-104 triangles_shifts */
                            0, sh_upright, sh_up, 0, sh_right, sh_upright};
    for (int y = 2; y < h - 2; y++) {
        for (int x = 1; x < w - 2; x++) {
            const long p = (long)y * w + x;
            if (pix_to_vert[p] == -1) continue;                                           /* :113-114 */
            int tr[4] = {0, 0, 0, 0};
            tr[0] = tri_ok(depth, p, p + sh_up, p + sh_right);                            /* :117 */
            tr[1] = tri_ok(depth, p + sh_right, p + sh_up, p + sh_upright);               /* :118 */
            if (!tr[0] && !tr[1]) {
                tr[2] = tri_ok(depth, p, p + sh_up, p + sh_upright);                      /* :122 */
                tr[3] = tri_ok(depth, p, p + sh_upright, p + sh_right);                   /* :123 */
            }
            for (int i = 0; i < 4; i++) {
                if (!tr[i]) continue;
                const int m1 = pix_to_vert[p + tshift[i * 3]];
                const int m2 = pix_to_vert[p + tshift[i * 3 + 1]];
                const int m3 = pix_to_vert[p + tshift[i * 3 + 2]];
                if (m1 == -1 || m2 == -1 || m3 == -1) continue;                           /* :133-134 */
                out[3 * n] = m1 + index_base;
                out[3 * n + 1] = m2 + index_base;
                out[3 * n + 2] = m3 + index_base;
                n++;
            }
        }
    }
    return n;
}

long orc_generate_mesh(int n_maps, const uint8_t *depth_maps, const uint8_t *depth_colors,
                       const int *widths, const int *heights, const float *intr, const float *wt,
                       const float *bounds6, orc_vertex *out, int *per_map_counts, int *out_tri, long *n_triangles)
{
    long dpos = 0, cpos = 0, nv = 0, nt = 0;
    long maxpix = 1;
    for (int i = 0; i < n_maps; i++) if ((long)widths[i] * heights[i] > maxpix) maxpix = (long)widths[i] * heights[i];
    int *p2v = (int *)malloc(sizeof(int) * (size_t)maxpix);
    for (int i = 0; i < n_maps; i++) {
        const long np = (long)widths[i] * heights[i];
        const uint16_t *d = (const uint16_t *)(depth_maps + dpos);
        const int c = orc_create_vertices(d, depth_colors + cpos, widths[i], heights[i], intr + 7 * i, wt + 12 * i, bounds6,
                                          out + nv, NULL, p2v);
        /* generateTriangles :1659-1691 on this sensor, then formMesh's index rebase :1614-1626 (act_vertices = nv) */
        nt += orc_generate_triangles(d, p2v, widths[i], heights[i], (int)nv, out_tri + 3 * nt);
        if (per_map_counts) per_map_counts[i] = c;
        nv += c;
        dpos += np * 2;
        cpos += np * 3;
    }
    free(p2v);
    if (n_triangles) *n_triangles = nt;
    return nv;
}

/* ------------------------------------------------------------------------------------------
 * exact nearest neighbour
 * ---------------------------------------------------------------------------------------- */

/* PointCloud::kdtree_distance, include/NativeUtils/icp.h:40-47 */
static inline float dist2(const float *q, const float *p)
{
    const float d0 = q[0] - p[0];
    const float d1 = q[1] - p[1];
    const float d2 = q[2] - p[2];
    return d0 * d0 + d1 * d1 + d2 * d2;
}

void orc_nn_brute(const float *targets, int n1, const float *queries, int n2,
                  int64_t *idx, float *dist, int n_threads)
{
#pragma omp parallel for schedule(static) num_threads(n_threads > 0 ? n_threads : 1)
    for (int i = 0; i < n2; i++) {
        const float *q = queries + 3 * (long)i;
        float best = INFINITY;
        int64_t bi = -1;
        for (int k = 0; k < n1; k++) {
            float d = dist2(q, targets + 3 * (long)k);
            if (d < best) { best = d; bi = k; }   /* strict: lowest index wins ties */
        }
        idx[i] = bi;
        dist[i] = best;
    }
}

/* A small exact kd-tree (median split on the widest dimension, leaves of <= 10 points like
 * nanoflann's default leaf size, include/nanoflann.h:408).  Search prunes with the single
 * splitting-plane bound only; with round-to-nearest monotonicity fl((q-s)^2) <= computed
 * dist2(q,p) for every p beyond the plane, so pruning is exact w.r.t. the f32 distances. */
typedef struct {
    int   left, right;     /* children, -1 for leaf */
    int   lo, hi;          /* leaf: range in perm   */
    int   dim;
    float split_lo, split_hi;   /* max coord of left subtree, min coord of right subtree along dim */
} kd_node;

typedef struct {
    const float *pts;
    int   *perm;
    kd_node *nodes;
    int   n_nodes, cap;
} kd_tree;

/* quickselect on perm[lo,hi) by coordinate dim so that perm[mid] is the median */
static void kd_select(const float *pts, int *perm, int lo, int hi, int mid, int dim)
{
    while (hi - lo > 1) {
        float pivot = pts[3 * (long)perm[lo + (hi - lo) / 2] + dim];
        int i = lo, j = hi - 1;
        while (i <= j) {
            while (pts[3 * (long)perm[i] + dim] < pivot) i++;
            while (pts[3 * (long)perm[j] + dim] > pivot) j--;
            if (i <= j) { int t = perm[i]; perm[i] = perm[j]; perm[j] = t; i++; j--; }
        }
        if (mid <= j) hi = j + 1;
        else if (mid >= i) lo = i;
        else break;
    }
}

static int kd_build(kd_tree *T, int lo, int hi)
{
    if (T->n_nodes == T->cap) {
        T->cap *= 2;
        T->nodes = (kd_node *)realloc(T->nodes, sizeof(kd_node) * (size_t)T->cap);
    }
    int id = T->n_nodes++;
    kd_node nd;
    nd.left = nd.right = -1; nd.lo = lo; nd.hi = hi; nd.dim = 0; nd.split_lo = nd.split_hi = 0;
    if (hi - lo > 10) {
        float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
        for (int i = lo; i < hi; i++)
            for (int d = 0; d < 3; d++) {
                float v = T->pts[3 * (long)T->perm[i] + d];
                if (v < mn[d]) mn[d] = v;
                if (v > mx[d]) mx[d] = v;
            }
        int dim = 0;
        if (mx[1] - mn[1] > mx[dim] - mn[dim]) dim = 1;
        if (mx[2] - mn[2] > mx[dim] - mn[dim]) dim = 2;
        if (mx[dim] > mn[dim]) {
            int mid = lo + (hi - lo) / 2;
            kd_select(T->pts, T->perm, lo, hi, mid, dim);
            float slo = -INFINITY, shi = INFINITY;
            for (int i = lo; i < mid; i++) { float v = T->pts[3 * (long)T->perm[i] + dim]; if (v > slo) slo = v; }
            for (int i = mid; i < hi; i++) { float v = T->pts[3 * (long)T->perm[i] + dim]; if (v < shi) shi = v; }
            nd.dim = dim; nd.split_lo = slo; nd.split_hi = shi;
            T->nodes[id] = nd;
            int l = kd_build(T, lo, mid);
            int r = kd_build(T, mid, hi);
            T->nodes[id].left = l;
            T->nodes[id].right = r;
            return id;
        }
    }
    T->nodes[id] = nd;
    return id;
}

static void kd_search(const kd_tree *T, int id, const float *q, float *best, int64_t *bi)
{
    const kd_node *nd = &T->nodes[id];
    if (nd->left < 0) {
        for (int i = nd->lo; i < nd->hi; i++) {
            int k = T->perm[i];
            float d = dist2(q, T->pts + 3 * (long)k);
            if (d < *best || (d == *best && k < *bi)) { *best = d; *bi = k; }
        }
        return;
    }
    float v = q[nd->dim];
    /* distance from q to each child's slab along dim (0 when inside) */
    float dl = v - nd->split_lo; if (dl < 0) dl = 0;    /* left child holds coords <= split_lo  */
    float dr = nd->split_hi - v; if (dr < 0) dr = 0;    /* right child holds coords >= split_hi */
    int first = nd->left, second = nd->right;
    float dsecond = dr;
    if (dr < dl) { first = nd->right; second = nd->left; dsecond = dl; }
    kd_search(T, first, q, best, bi);
    if (dsecond * dsecond <= *best)            /* <= keeps equal-distance lower indices reachable */
        kd_search(T, second, q, best, bi);
}

void orc_nn_kdtree(const float *targets, int n1, const float *queries, int n2,
                   int64_t *idx, float *dist, int n_threads)
{
    kd_tree T;
    T.pts = targets;
    T.perm = (int *)malloc(sizeof(int) * (size_t)(n1 > 0 ? n1 : 1));
    for (int i = 0; i < n1; i++) T.perm[i] = i;
    T.cap = 64 + n1 / 4;
    T.nodes = (kd_node *)malloc(sizeof(kd_node) * (size_t)T.cap);
    T.n_nodes = 0;
    if (n1 > 0) kd_build(&T, 0, n1);
#pragma omp parallel for schedule(dynamic, 256) num_threads(n_threads > 0 ? n_threads : 1)
    for (int i = 0; i < n2; i++) {
        float best = INFINITY;
        int64_t bi = -1;
        if (n1 > 0) kd_search(&T, 0, queries + 3 * (long)i, &best, &bi);
        idx[i] = bi;
        dist[i] = best;
    }
    free(T.nodes);
    free(T.perm);
}

/* ------------------------------------------------------------------------------------------
 * ICP
 * ---------------------------------------------------------------------------------------- */

/* 3x3 SVD by one-sided Jacobi in double, then Rn = U*Vt with the det fix-up.
 * icp.cpp:152-163 uses OpenCV 3.2.0 cv::SVD (f32 Jacobi, library binary absent): U*Vt is the
 * orthogonal polar factor of M, which does not depend on the SVD algorithm beyond rounding. */
static void svd3(const double A[9], double U[9], double w[3], double V[9])
{
    /* columns of B = A*V are rotated until mutually orthogonal */
    double B[9];
    memcpy(B, A, sizeof(B));
    for (int i = 0; i < 9; i++) V[i] = (i % 4 == 0) ? 1.0 : 0.0;
    for (int sweep = 0; sweep < 60; sweep++) {
        int rotations = 0;
        for (int p = 0; p < 2; p++)
            for (int q = p + 1; q < 3; q++) {
                double a = 0, b = 0, c = 0;
                for (int k = 0; k < 3; k++) {
                    a += B[3 * k + p] * B[3 * k + p];
                    b += B[3 * k + q] * B[3 * k + q];
                    c += B[3 * k + p] * B[3 * k + q];
                }
                if (fabs(c) <= 1e-300 || c * c <= 1e-32 * (a * b)) continue;  /* columns orthogonal to ~1e-16 */
                rotations++;
                double zeta = (b - a) / (2.0 * c);
                double tt = (zeta >= 0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
                double cs = 1.0 / sqrt(1.0 + tt * tt), sn = cs * tt;
                for (int k = 0; k < 3; k++) {
                    double bp = B[3 * k + p], bq = B[3 * k + q];
                    B[3 * k + p] = cs * bp - sn * bq;
                    B[3 * k + q] = sn * bp + cs * bq;
                    double vp = V[3 * k + p], vq = V[3 * k + q];
                    V[3 * k + p] = cs * vp - sn * vq;
                    V[3 * k + q] = sn * vp + cs * vq;
                }
            }
        if (rotations == 0) break;
    }
    for (int j = 0; j < 3; j++) {
        double s = 0;
        for (int k = 0; k < 3; k++) s += B[3 * k + j] * B[3 * k + j];
        w[j] = sqrt(s);
    }
    /* sort singular values descending (cv::SVD convention) */
    int ord[3] = {0, 1, 2};
    for (int i = 0; i < 2; i++)
        for (int j = i + 1; j < 3; j++)
            if (w[ord[j]] > w[ord[i]]) { int t = ord[i]; ord[i] = ord[j]; ord[j] = t; }
    double Bs[9], Vs[9], ws[3];
    for (int j = 0; j < 3; j++) {
        ws[j] = w[ord[j]];
        for (int k = 0; k < 3; k++) { Bs[3 * k + j] = B[3 * k + ord[j]]; Vs[3 * k + j] = V[3 * k + ord[j]]; }
    }
    memcpy(V, Vs, sizeof(Vs));
    memcpy(w, ws, sizeof(ws));
    /* U columns = B columns / w; rank-deficient columns completed by cross products */
    for (int j = 0; j < 3; j++)
        for (int k = 0; k < 3; k++)
            U[3 * k + j] = (w[j] > 1e-300) ? Bs[3 * k + j] / w[j] : 0.0;
    if (!(w[0] > 1e-300)) { for (int i = 0; i < 9; i++) U[i] = (i % 4 == 0) ? 1.0 : 0.0; return; }
    if (!(w[1] > 1e-12 * w[0])) {
        /* pick any unit vector orthogonal to u0 */
        double u0[3] = {U[0], U[3], U[6]};
        int m = 0;
        if (fabs(u0[1]) < fabs(u0[m])) m = 1;
        if (fabs(u0[2]) < fabs(u0[m])) m = 2;
        double e[3] = {0, 0, 0}; e[m] = 1;
        double d = u0[m];
        double v[3] = {e[0] - d * u0[0], e[1] - d * u0[1], e[2] - d * u0[2]};
        double nv = sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
        U[1] = v[0] / nv; U[4] = v[1] / nv; U[7] = v[2] / nv;
    }
    if (!(w[2] > 1e-12 * w[0])) {
        /* u2 = +-(u0 x u1); sign chosen so that det(U) = det(V) (then det(U*Vt) = +1, the proper polar limit) */
        double a0 = U[0], a1 = U[3], a2 = U[6], b0 = U[1], b1 = U[4], b2 = U[7];
        double c0 = a1 * b2 - a2 * b1, c1 = a2 * b0 - a0 * b2, c2 = a0 * b1 - a1 * b0;
        double detV = V[0] * (V[4] * V[8] - V[5] * V[7]) - V[1] * (V[3] * V[8] - V[5] * V[6]) + V[2] * (V[3] * V[7] - V[4] * V[6]);
        double s = detV < 0 ? -1.0 : 1.0;
        U[2] = s * c0; U[5] = s * c1; U[8] = s * c2;
    }
}

void orc_kabsch_rotation(const float *M9, float *Rn9)
{
    double A[9], U[9], w[3], V[9];
    for (int i = 0; i < 9; i++) A[i] = M9[i];
    svd3(A, U, w, V);
    /* svd.u, svd.vt are CV_32F in the reference: round the factors first, then the small f32 products */
    float Uf[9], Vtf[9];
    for (int r = 0; r < 3; r++)
        for (int c = 0; c < 3; c++) { Uf[3 * r + c] = (float)U[3 * r + c]; Vtf[3 * r + c] = (float)V[3 * c + r]; }
    float Rn[9];
    for (int r = 0; r < 3; r++)
        for (int c = 0; c < 3; c++)
            Rn[3 * r + c] = Uf[3 * r] * Vtf[c] + Uf[3 * r + 1] * Vtf[3 + c] + Uf[3 * r + 2] * Vtf[6 + c];
    /* cv::determinant of a 3x3 CV_32F is evaluated in double (icp.cpp:157) */
    double det = (double)Rn[0] * ((double)Rn[4] * Rn[8] - (double)Rn[5] * Rn[7])
               - (double)Rn[1] * ((double)Rn[3] * Rn[8] - (double)Rn[5] * Rn[6])
               + (double)Rn[2] * ((double)Rn[3] * Rn[7] - (double)Rn[4] * Rn[6]);
    if (det < 0) {
        /* tempR = u * diag(1,1,-1) * vt  (icp.cpp:160-162) */
        for (int r = 0; r < 3; r++) Uf[3 * r + 2] = -Uf[3 * r + 2];
        for (int r = 0; r < 3; r++)
            for (int c = 0; c < 3; c++)
                Rn[3 * r + c] = Uf[3 * r] * Vtf[c] + Uf[3 * r + 1] * Vtf[3 + c] + Uf[3 * r + 2] * Vtf[6 + c];
    }
    memcpy(Rn9, Rn, sizeof(Rn));
}

float orc_icp(const float *verts1, float *verts2, int n1, int n2,
              float *R, float *t, int maxIter, int nn_mode, int n_threads,
              orc_icp_iter *trace)
{
    float error = 1;                                                     /* icp.cpp:85 */
    if (n1 <= 0 || n2 <= 0) return error;   /* nanoflann would throw on an empty cloud (nanoflann.h:904); callers guard */

    int64_t *indices = (int64_t *)malloc(sizeof(int64_t) * (size_t)n2);
    float *distances = (float *)malloc(sizeof(float) * (size_t)n2);
    int *matchIdxs = (int *)malloc(sizeof(int) * (size_t)n1);
    int cap = n1 < n2 ? n1 : n2;
    float *matched1 = (float *)malloc(sizeof(float) * 3 * (size_t)cap);
    float *matched2 = (float *)malloc(sizeof(float) * 3 * (size_t)cap);
    float *matchDistances = (float *)malloc(sizeof(float) * (size_t)cap);

    for (int iter = 0; iter < maxIter; iter++) {                         /* :87 */
        /* :91-93 FindClosestPointForEach (tree over cloud1 = verts1 copy, queries = verts2) */
        if (nn_mode == 0) orc_nn_brute(verts1, n1, verts2, n2, indices, distances, n_threads);
        else              orc_nn_kdtree(verts1, n1, verts2, n2, indices, distances, n_threads);

        /* :95-126 one-to-one matching, i ascending, later i wins ties */
        int m = 0;
        for (int k = 0; k < n1; k++) matchIdxs[k] = -1;
        for (int i = 0; i < n2; i++) {
            int pos = matchIdxs[indices[i]];
            if (pos != -1) {
                if (matchDistances[pos] < distances[i]) continue;
            }
            if (pos == -1) {
                memcpy(matched1 + 3 * (long)m, verts1 + 3 * indices[i], 3 * sizeof(float));
                memcpy(matched2 + 3 * (long)m, verts2 + 3 * (long)i, 3 * sizeof(float));
                matchDistances[m] = distances[i];
                matchIdxs[indices[i]] = m;
                m++;
            } else {
                memcpy(matched2 + 3 * (long)pos, verts2 + 3 * (long)i, 3 * sizeof(float));
                matchDistances[pos] = distances[i];
            }
        }

        /* :128 RejectOutlierMatches(.., 2.5) -> :34-54 GetStandardDeviation: sequential f32 sums,
         * pow(float,int) evaluates in double and is truncated back into the float accumulator. */
        float mean = 0;
        for (int i = 0; i < m; i++) mean += matchDistances[i];
        mean /= (float)(size_t)m;
        float sd = 0;
        for (int i = 0; i < m; i++) sd = (float)((double)sd + pow((double)(matchDistances[i] - mean), 2));
        sd /= (float)(size_t)m;
        sd = sqrtf(sd);
        const float maxStdDev = 2.5f;
        int mk = 0;
        for (int i = 0; i < m; i++) {                                     /* :62-69 order preserved */
            if (matchDistances[i] > maxStdDev * sd) continue;
            if (mk != i) {
                memcpy(matched1 + 3 * (long)mk, matched1 + 3 * (long)i, 3 * sizeof(float));
                memcpy(matched2 + 3 * (long)mk, matched2 + 3 * (long)i, 3 * sizeof(float));
            }
            mk++;
        }

        if (mk == 0) break;   /* cv::reduce on an empty matrix would throw across the ABI; stop instead */

        /* :141 cv::reduce(matched1 - matched2, tempT, 0, CV_REDUCE_AVG): f32 element-wise difference,
         * f32 column sums top to bottom, then scaled by (float)(1.0/rows). */
        float T[3] = {0, 0, 0};
        for (int j = 0; j < mk; j++)
            for (int c = 0; c < 3; c++) {
                float d = matched1[3 * (long)j + c] - matched2[3 * (long)j + c];
                T[c] = T[c] + d;
            }
        {
            float scale = (float)(1.0 / (double)mk);
            for (int c = 0; c < 3; c++) T[c] = T[c] * scale;
        }

        /* :143-150 translate all of verts2 and the matched source points */
        for (int i = 0; i < n2; i++)
            for (int c = 0; c < 3; c++) verts2[3 * (long)i + c] += T[c];
        for (int j = 0; j < mk; j++)
            for (int c = 0; c < 3; c++) matched2[3 * (long)j + c] += T[c];

        /* :152 M = matched2^T * matched1 (3 x m' times m' x 3; OpenCV's f32 gemm accumulates in double) */
        float M[9];
        {
            double acc[9] = {0};
            for (int j = 0; j < mk; j++)
                for (int a = 0; a < 3; a++)
                    for (int b = 0; b < 3; b++)
                        acc[3 * a + b] += (double)matched2[3 * (long)j + a] * (double)matched1[3 * (long)j + b];
            for (int i = 0; i < 9; i++) M[i] = (float)acc[i];
        }

        /* :153-163 */
        float Rn[9];
        orc_kabsch_rotation(M, Rn);

        /* :165 verts2Mat = verts2Mat * tempR (row vectors) */
        for (int i = 0; i < n2; i++) {
            float *v = verts2 + 3 * (long)i;
            float x = v[0], y = v[1], z = v[2];
            v[0] = x * Rn[0] + y * Rn[3] + z * Rn[6];
            v[1] = x * Rn[1] + y * Rn[4] + z * Rn[7];
            v[2] = x * Rn[2] + y * Rn[5] + z * Rn[8];
        }
        /* :167 matT += tempT * matR.t()   (R before the update) */
        {
            float add[3];
            for (int c = 0; c < 3; c++) add[c] = T[0] * R[3 * c] + T[1] * R[3 * c + 1] + T[2] * R[3 * c + 2];
            for (int c = 0; c < 3; c++) t[c] += add[c];
        }
        /* :168 matR = matR * tempR */
        {
            float Rnew[9];
            for (int r = 0; r < 3; r++)
                for (int c = 0; c < 3; c++)
                    Rnew[3 * r + c] = R[3 * r] * Rn[c] + R[3 * r + 1] * Rn[3 + c] + R[3 * r + 2] * Rn[6 + c];
            memcpy(R, Rnew, sizeof(Rnew));
        }
        if (trace) {
            trace[iter].n_matched = m;
            trace[iter].n_kept = mk;
            trace[iter].mean = mean;
            trace[iter].stddev = sd;
            memcpy(trace[iter].T, T, sizeof(T));
            memcpy(trace[iter].Rn, Rn, sizeof(Rn));
        }
    }
    free(matchDistances); free(matched2); free(matched1); free(matchIdxs); free(distances); free(indices);
    return error;                                                        /* :176 */
}

/* refineWorker_DoWork, LiveScanServer/MainWindowForm.cs:330-410 */
void orc_refine(int n_sensors, float **clouds, const int *n, int n_refine_iters, int n_icp_iters,
                float *world_R, float *world_t, float *Rs_out, float *Ts_out,
                int nn_mode, int n_threads)
{
    float *Rs = (float *)calloc((size_t)n_sensors * 9, sizeof(float));
    float *Ts = (float *)calloc((size_t)n_sensors * 3, sizeof(float));
    for (int i = 0; i < n_sensors; i++)                                   /* :330-344 */
        for (int j = 0; j < 3; j++) Rs[9 * i + j + j * 3] = 1;

    long total = 0;
    for (int i = 0; i < n_sensors; i++) total += n[i];
    float *others = (float *)malloc(sizeof(float) * 3 * (size_t)(total > 0 ? total : 1));

    for (int it = 0; it < n_refine_iters; it++) {                         /* :347 */
        for (int i = 0; i < n_sensors; i++) {                             /* :349 */
            long n_others = 0;
            for (int j = 0; j < n_sensors; j++) {                         /* :352-357 */
                if (j == i) continue;
                memcpy(others + 3 * n_others, clouds[j], sizeof(float) * 3 * (size_t)n[j]);
                n_others += n[j];
            }
            orc_icp(others, clouds[i], (int)n_others, n[i], Rs + 9 * i, Ts + 3 * i,
                    n_icp_iters, nn_mode, n_threads, NULL);               /* :370 */
        }
    }

    /* :382-410 pose composition.  The reference writes worldTransforms[i].R[j,k] inside the
     * k-loop while later rows still read worldTransforms[i].R[l,k]; that aliasing is kept. */
    if (world_R && world_t) {
        for (int i = 0; i < n_sensors; i++) {
            float *WR = world_R + 9 * i, *Wt = world_t + 3 * i;
            const float *Ri = Rs + 9 * i, *Ti = Ts + 3 * i;
            float tempT[3] = {0, 0, 0};
            float tempR[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
            for (int j = 0; j < 3; j++) {
                for (int k = 0; k < 3; k++) tempT[j] += Ti[k] * WR[3 * k + j];
                Wt[j] += tempT[j];
            }
            for (int j = 0; j < 3; j++)
                for (int k = 0; k < 3; k++) {
                    for (int l = 0; l < 3; l++) tempR[3 * j + k] += Ri[l * 3 + j] * WR[3 * l + k];
                    WR[3 * j + k] = tempR[3 * j + k];
                }
        }
    }
    if (Rs_out) memcpy(Rs_out, Rs, sizeof(float) * 9 * (size_t)n_sensors);
    if (Ts_out) memcpy(Ts_out, Ts, sizeof(float) * 3 * (size_t)n_sensors);
    free(others); free(Ts); free(Rs);
}

/* ======================================================================================================
 * Wire / disk formats
 * ====================================================================================================== */

/* TransferServer.cs:177-201 (no triangles) and :203-270 (mesh) */
int orc_form_chunks(const orc_vertex *v, int nv, const int *tri, int nt, orc_vertex *new_v, int *new_tri,
                    int *v_chunks, int *t_chunks, int *n_new_v)
{
    const int limit = ORC_CHUNK_LIMIT;
    int n_chunks = 0;
    if (nt <= 0) {                                   /* formVerticesChunks :177-201 */
        int cur = 0;
        while (cur < nv) {
            int size = nv - cur < limit ? nv - cur : limit;
            v_chunks[n_chunks] = size;
            t_chunks[n_chunks] = 0;
            n_chunks++;
            cur += size;
        }
        memcpy(new_v, v, (size_t)nv * sizeof(orc_vertex));
        *n_new_v = nv;
        return n_chunks;
    }
    int *chunk_index = (int *)malloc((size_t)(nv > 0 ? nv : 1) * sizeof(int));
    int *vertices_map = (int *)malloc((size_t)(nv > 0 ? nv : 1) * sizeof(int));
    for (int i = 0; i < nv; i++) chunk_index[i] = -1;                          /* :221-222 */
    int triangles_chunk_start = 0, current_chunk = 0, current_vertex = 0, in_chunk = 0;
    for (int t = 0; t < nt * 3; t++) {                                          /* :230-254 */
        const int val = tri[t];
        if (chunk_index[val] != current_chunk) {
            new_v[current_vertex] = v[val];
            vertices_map[val] = in_chunk;
            chunk_index[val] = current_chunk;
            current_vertex++;
            new_tri[t] = in_chunk;
            in_chunk++;
        } else {
            new_tri[t] = vertices_map[val];
        }
        if (in_chunk >= limit && ((t + 1) % 3) == 0) {
            current_chunk++;
            v_chunks[n_chunks] = in_chunk;
            t_chunks[n_chunks] = (t - triangles_chunk_start) / 3;
            n_chunks++;
            in_chunk = 0;
            triangles_chunk_start = t;                                          /* sic: t, not t + 1 */
        }
    }
    if (in_chunk != 0) {                                                        /* :256-260 */
        v_chunks[n_chunks] = in_chunk;
        t_chunks[n_chunks] = (nt * 3 - triangles_chunk_start) / 3;
        n_chunks++;
    }
    free(chunk_index);
    free(vertices_map);
    *n_new_v = current_vertex;
    return n_chunks;
}

static void put_i32(uint8_t *p, int v) { memcpy(p, &v, 4); }

/* TransferSocket.cs:50-104 */
long orc_transfer_frame(const orc_vertex *v, int nv, const int *tri, int nt, uint8_t *out, long cap)
{
    const long max_v = nt > 0 ? 3L * nt : nv;
    const long max_chunks = 3L * (nt > 0 ? nt : 0) / ORC_CHUNK_LIMIT + nv / ORC_CHUNK_LIMIT + 2;
    orc_vertex *nvv = (orc_vertex *)malloc((size_t)(max_v > 0 ? max_v : 1) * sizeof(orc_vertex));
    int *ntri = (int *)malloc((size_t)(nt > 0 ? 3L * nt : 1) * sizeof(int));
    int *vc = (int *)malloc((size_t)max_chunks * sizeof(int)), *tc = (int *)malloc((size_t)max_chunks * sizeof(int));
    int n_send = 0;
    const int n_chunks = orc_form_chunks(v, nv, tri, nt, nvv, ntri, vc, tc, &n_send);
    const int n_tri = nt > 0 ? nt : 0;
    const long need = 12 + 8L * n_chunks + 15L * n_send + 12L * n_tri;
    long ret = need;
    if (need > cap) {
        ret = -need;
    } else {
        uint8_t *p = out;
        put_i32(p, n_send); put_i32(p + 4, n_tri); put_i32(p + 8, n_chunks);   /* :92-94 */
        p += 12;
        memcpy(p, vc, 4L * n_chunks); p += 4L * n_chunks;                       /* :95 */
        memcpy(p, tc, 4L * n_chunks); p += 4L * n_chunks;                       /* :96 */
        for (int i = 0; i < n_send; i++) {                                      /* :97, packed at :66-75 */
            memcpy(p + 12L * i, &nvv[i].X, 4); memcpy(p + 12L * i + 4, &nvv[i].Y, 4); memcpy(p + 12L * i + 8, &nvv[i].Z, 4);
        }
        p += 12L * n_send;
        for (int i = 0; i < n_send; i++) { p[3L * i] = nvv[i].R; p[3L * i + 1] = nvv[i].G; p[3L * i + 2] = nvv[i].B; }   /* :98 */
        p += 3L * n_send;
        if (n_tri) memcpy(p, ntri, 12L * n_tri);                                /* :99 */
    }
    free(nvv); free(ntri); free(vc); free(tc);
    return ret;
}

/* Utils.cs:222-262 (binary branch) */
long orc_ply_binary(const orc_vertex *v, int nv, const int *tri, int nt, uint8_t *out, long cap)
{
    char hdr[512];
    const int hl = snprintf(hdr, sizeof hdr,
                            "ply\nformat binary_little_endian 1.0\r\n"                       /* WriteLine :234 */
                            "element vertex %d\n"
                            "property float x\nproperty float y\nproperty float z\nproperty uchar red\nproperty uchar green\nproperty uchar blue\n"
                            "element face %d\n"
                            "property list uchar int vertex_index\n"
                            "end_header\n", nv, nt);
    const long need = hl + 15L * nv + 13L * nt;
    if (need > cap) return -need;
    uint8_t *p = out;
    memcpy(p, hdr, (size_t)hl); p += hl;
    for (int j = 0; j < nv; j++) {                                              /* :246-255 */
        memcpy(p, &v[j].X, 4); memcpy(p + 4, &v[j].Y, 4); memcpy(p + 8, &v[j].Z, 4);
        p[12] = v[j].R; p[13] = v[j].G; p[14] = v[j].B;
        p += 15;
    }
    for (int j = 0; j < nt; j++) {                                              /* :257-263 */
        p[0] = 3;
        memcpy(p + 1, tri + 3L * j, 12);
        p += 13;
    }
    return need;
}

/* walks the serialized body block (liveScanClient.cpp:233-268 / KinectSocket.cs:262-303); returns its length or -1 */
static long body_block_length(const uint8_t *b, long avail, int *n_bodies)
{
    if (avail < 4) return -1;
    int nb;
    memcpy(&nb, b, 4);
    if (nb < 0) return -1;
    long pos = 4;
    for (int i = 0; i < nb; i++) {
        if (pos + 5 > avail) return -1;
        int nj;
        memcpy(&nj, b + pos + 1, 4);
        if (nj < 0) return -1;
        pos += 5 + 28L * nj;
        if (pos > avail) return -1;
    }
    *n_bodies = nb;
    return pos;
}

long orc_frame_encode(const uint8_t *depth, const uint8_t *rgb, int w, int h, const uint8_t *bodies, int bodies_bytes,
                      uint8_t *out, long cap)
{
    const long P = (long)w * h;
    static const uint8_t no_bodies[4] = {0, 0, 0, 0};
    if (!bodies || bodies_bytes < 4) { bodies = no_bodies; bodies_bytes = 4; }
    const long size = 5 * P + bodies_bytes;
    if (16 + size > cap) return -(16 + size);
    const int isize = (int)size, comp = 0;
    memcpy(out, &isize, 4); memcpy(out + 4, &comp, 4); memcpy(out + 8, &w, 4); memcpy(out + 12, &h, 4);   /* :281-286 */
    memcpy(out + 16, depth, (size_t)(2 * P));                                   /* :211 */
    memcpy(out + 16 + 2 * P, rgb, (size_t)(3 * P));                             /* :214-231 */
    memcpy(out + 16 + 5 * P, bodies, (size_t)bodies_bytes);                     /* :233-268 */
    return 16 + size;
}

int orc_frame_decode(const uint8_t *msg, long len, int *w, int *h, const uint8_t **depth, const uint8_t **rgb,
                     const uint8_t **bodies, int *bodies_bytes, int *n_bodies)
{
    if (len < 16) return -1;
    int size, comp, ww, hh;
    memcpy(&size, msg, 4); memcpy(&comp, msg + 4, 4); memcpy(&ww, msg + 8, 4); memcpy(&hh, msg + 12, 4);   /* KinectSocket.cs:229-241 */
    if (size <= 0 || comp != 0 || ww < 0 || hh < 0 || 16 + (long)size > len) return -1;
    const long P = (long)ww * hh;
    if (5 * P + 4 > size) return -1;
    int nb = 0;
    const long bl = body_block_length(msg + 16 + 5 * P, size - 5 * P, &nb);
    if (bl < 0) return -1;
    *w = ww; *h = hh;
    *depth = msg + 16; *rgb = msg + 16 + 2 * P; *bodies = msg + 16 + 5 * P;
    *bodies_bytes = (int)(size - 5 * P);
    *n_bodies = nb;
    return 0;
}

long orc_recording_append(uint8_t *out, long cap, const uint8_t *frame, int len, int timestamp_ms)
{
    char hdr[96];
    const int hl = snprintf(hdr, sizeof hdr, "bufferSize= %d\nframe_timestamp= %d\n", len, timestamp_ms);   /* :123 */
    const long need = hl + (long)len + 1;
    if (need > cap) return -need;
    memcpy(out, hdr, (size_t)hl);
    if (len > 0) memcpy(out + hl, frame, (size_t)len);
    out[hl + len] = '\n';                                                       /* :127 */
    return need;
}

/* fscanf("%s %d %s %d") as the reader uses it (:66): skip white space, token, int, token, int */
static long scan_token(const uint8_t *f, long len, long pos, long *end)
{
    while (pos < len && (f[pos] == ' ' || f[pos] == '\n' || f[pos] == '\r' || f[pos] == '\t')) pos++;
    if (pos >= len) return -1;
    long e = pos;
    while (e < len && !(f[e] == ' ' || f[e] == '\n' || f[e] == '\r' || f[e] == '\t')) e++;
    *end = e;
    return pos;
}

static int scan_int(const uint8_t *f, long len, long *pos, int *out)
{
    long e, b = scan_token(f, len, *pos, &e);
    if (b < 0 || e - b > 11) return -1;
    char tmp[16];
    memcpy(tmp, f + b, (size_t)(e - b));
    tmp[e - b] = 0;
    char *endp;
    const long v = strtol(tmp, &endp, 10);
    if (*endp != 0 || endp == tmp) return -1;
    *out = (int)v;
    *pos = e;
    return 0;
}

long orc_recording_next(const uint8_t *file, long len, long pos, long *frame_off, int *frame_len, int *timestamp_ms)
{
    long e;
    int size, ts;
    if (scan_token(file, len, pos, &e) < 0) return -1;                           /* "bufferSize=" */
    pos = e;
    if (scan_int(file, len, &pos, &size) < 0) return -1;
    if (scan_token(file, len, pos, &e) < 0) return -1;                           /* "frame_timestamp=" */
    pos = e;
    if (scan_int(file, len, &pos, &ts) < 0) return -1;
    if (size < 0) return -1;
    *frame_len = size;
    *timestamp_ms = ts;
    if (size == 0) { *frame_off = pos; return pos; }                             /* :72-73: returns before the fgetc */
    pos += 1;                                                                    /* fgetc '\n' :75 */
    if (pos + size > len) return -1;
    *frame_off = pos;
    pos += size;
    if (pos < len) pos += 1;                                                     /* fgetc '\n' :78 */
    return pos;
}
