// ref_abi_harness.cpp -- pins the drop-in boundary against the REFERENCE's own header.
//
// TEST INFRASTRUCTURE ONLY.  Compiled by oracle/Makefile (targets _ref/ref_abi, _ref/libref_abi.so) against
// include/NativeUtils/depthprocessing.h and include/NativeUtils/icp.h where they lie under /root/reference; nothing of the
// reference is copied here.  Recipe-level accommodations, the same as for ref_nn: -D'__declspec(x)=' -D__stdcall= for the MSVC
// keywords, -include cstring / string for the memcpy and std::string the header uses without including them.
//
// What it pins (b):
//   * sizeof / offsetof of VertexC4ubV3f, Mesh, Point3f as the reference's compiler lays them out (depthprocessing.h:29-33,42-48,
//     icp.h:15-18);
//   * what IntrinsicCameraParameters(float*) and WorldTranformation(float*) (+ inv()) make of the caller's 7 / 12 floats
//     (depthprocessing.h:50-78,91-98) -- the unpack order this library's SensorParams packing has to reproduce;
//   * the PROTOTYPES: this repository's include/NativeUtils.h is included below inside a namespace, and every export the two
//     headers share must have the same type, argument for argument (depthprocessing.h:103-112, icp.h:65).  A mismatch does not
//     compile (the static_asserts below).
//
// ref_abi prints the values as JSON (tests/golden/make_abi_golden.py stores them as tests/golden/abi_reference.json);
// libref_abi.so exports the same through ref_abi_layout / ref_unpack_intrinsics / ref_unpack_world for the live check.
#include "NativeUtils/depthprocessing.h"

#include <cstddef>
#include <cstdio>
#include <type_traits>

// this library's header: structs kept apart by a namespace, the shared export names by a prefix (two extern "C" declarations of
// one name must agree exactly, and these take the Mesh / Point3f of their own header)
#include <stdbool.h>
#include <stdint.h>
#include <tuple>
#define generateVerticesFromDepthMap ours_generateVerticesFromDepthMap
#define generateMeshFromDepthMaps ours_generateMeshFromDepthMaps
#define depthMapAndColorSetRadialCorrection ours_depthMapAndColorSetRadialCorrection
#define deleteMesh ours_deleteMesh
#define createMesh ours_createMesh
#define ICP ours_ICP
namespace ours {
#include "NativeUtils.h"
}
#undef generateVerticesFromDepthMap
#undef generateMeshFromDepthMaps
#undef depthMapAndColorSetRadialCorrection
#undef deleteMesh
#undef createMesh
#undef ICP

// The exports take Mesh* / Point3f* of their own header, so the function types cannot be compared with is_same directly; what
// can be compared is everything else: argument count, every scalar / pointer argument, and the layout of the two structs.
template <class F>
struct sig;
template <class R, class... A>
struct sig<R (*)(A...)> {
    static constexpr int n = sizeof...(A);
    typedef R ret;
    template <int I>
    using arg = typename std::tuple_element<I, std::tuple<A...>>::type;
};

template <class Fr, class Fo, int I>
constexpr bool same_arg()
{
    typedef typename sig<Fr>::template arg<I> ar;
    typedef typename sig<Fo>::template arg<I> ao;
    // struct pointers: Mesh* against ours::Mesh*, Point3f* against ours::Point3f*
    return std::is_same<ar, ao>::value || (std::is_same<ar, ::Mesh *>::value && std::is_same<ao, ours::Mesh *>::value) ||
           (std::is_same<ar, ::Point3f *>::value && std::is_same<ao, ours::Point3f *>::value);
}
template <class Fr, class Fo, int... I>
constexpr bool same_args(std::integer_sequence<int, I...>)
{
    return (same_arg<Fr, Fo, I>() && ...);
}
template <class Fr, class Fo>
constexpr bool same_signature()
{
    return sig<Fr>::n == sig<Fo>::n && std::is_same<typename sig<Fr>::ret, typename sig<Fo>::ret>::value &&
           same_args<Fr, Fo>(std::make_integer_sequence<int, sig<Fr>::n>());
}

#define LSN_SAME(name)                                                                                   \
    static_assert(same_signature<decltype(&::name), decltype(&ours::ours_##name)>(),                    \
                  #name ": include/NativeUtils.h does not declare the reference's prototype")
LSN_SAME(generateVerticesFromDepthMap);          // depthprocessing.h:103-105
LSN_SAME(generateMeshFromDepthMaps);             // depthprocessing.h:108-110
LSN_SAME(depthMapAndColorSetRadialCorrection);   // depthprocessing.h:111
LSN_SAME(deleteMesh);                            // depthprocessing.h:112
LSN_SAME(ICP);                                   // icp.h:65

static_assert(sizeof(::VertexC4ubV3f) == sizeof(ours::VertexC4ubV3f) && offsetof(::VertexC4ubV3f, A) == offsetof(ours::VertexC4ubV3f, A) &&
                  offsetof(::VertexC4ubV3f, X) == offsetof(ours::VertexC4ubV3f, X) && offsetof(::VertexC4ubV3f, Z) == offsetof(ours::VertexC4ubV3f, Z),
              "VertexC4ubV3f layout");
static_assert(sizeof(::Mesh) == sizeof(ours::Mesh) && offsetof(::Mesh, vertices) == offsetof(ours::Mesh, vertices) &&
                  offsetof(::Mesh, nTriangles) == offsetof(ours::Mesh, nTriangles) && offsetof(::Mesh, triangles) == offsetof(ours::Mesh, triangles),
              "Mesh layout");
static_assert(sizeof(::Point3f) == sizeof(ours::Point3f) && offsetof(::Point3f, Z) == offsetof(ours::Point3f, Z), "Point3f layout");

// layout[0..7]  VertexC4ubV3f: sizeof, offsets of R G B A X Y Z
// layout[8..12] Mesh: sizeof, offsets of nVertices vertices nTriangles triangles
// layout[13..16] Point3f: sizeof, offsets of X Y Z
extern "C" int ref_abi_layout(int *out, int cap)
{
    const int v[] = {(int)sizeof(::VertexC4ubV3f), (int)offsetof(::VertexC4ubV3f, R), (int)offsetof(::VertexC4ubV3f, G), (int)offsetof(::VertexC4ubV3f, B),
                     (int)offsetof(::VertexC4ubV3f, A), (int)offsetof(::VertexC4ubV3f, X), (int)offsetof(::VertexC4ubV3f, Y), (int)offsetof(::VertexC4ubV3f, Z),
                     (int)sizeof(::Mesh), (int)offsetof(::Mesh, nVertices), (int)offsetof(::Mesh, vertices), (int)offsetof(::Mesh, nTriangles),
                     (int)offsetof(::Mesh, triangles),
                     (int)sizeof(::Point3f), (int)offsetof(::Point3f, X), (int)offsetof(::Point3f, Y), (int)offsetof(::Point3f, Z)};
    const int n = (int)(sizeof(v) / sizeof(v[0]));
    for (int i = 0; i < n && i < cap; i++) out[i] = v[i];
    return n;
}

// the reference's own constructor on the caller's 7 floats -> {cx, cy, fx, fy, r2, r4, r6} as the struct's members read
extern "C" void ref_unpack_intrinsics(float *p7, float *out7)
{
    IntrinsicCameraParameters k(p7);
    out7[0] = k.cx; out7[1] = k.cy; out7[2] = k.fx; out7[3] = k.fy; out7[4] = k.r2; out7[5] = k.r4; out7[6] = k.r6;
}

// the reference's own constructor on the caller's 12 floats (and optionally inv()) -> t[3], R[3][3] row-major
extern "C" void ref_unpack_world(float *p12, int inverse, float *t3, float *R9)
{
    WorldTranformation w(p12);
    if (inverse) w.inv();
    for (int i = 0; i < 3; i++) {
        t3[i] = w.t[i];
        for (int j = 0; j < 3; j++) R9[3 * i + j] = w.R[i][j];
    }
}

#ifndef REF_ABI_NO_MAIN
int main()
{
    int lay[32];
    const int n = ref_abi_layout(lay, 32);
    float p7[7], o7[7], p12[12], t[3], R[9], ti[3], Ri[9];
    for (int i = 0; i < 7; i++) p7[i] = 100.0f + (float)i;       // position-coded: the value tells where it came from
    for (int i = 0; i < 12; i++) p12[i] = 200.0f + (float)i;
    ref_unpack_intrinsics(p7, o7);
    ref_unpack_world(p12, 0, t, R);
    ref_unpack_world(p12, 1, ti, Ri);
    printf("{\n \"layout\": [");
    for (int i = 0; i < n; i++) printf("%s%d", i ? ", " : "", lay[i]);
    printf("],\n \"layout_fields\": \"VertexC4ubV3f: sizeof, offsetof R G B A X Y Z; Mesh: sizeof, offsetof nVertices vertices nTriangles triangles; Point3f: sizeof, offsetof X Y Z\",\n");
    printf(" \"intr_in\": [");
    for (int i = 0; i < 7; i++) printf("%s%g", i ? ", " : "", p7[i]);
    printf("],\n \"intr_members_cx_cy_fx_fy_r2_r4_r6\": [");
    for (int i = 0; i < 7; i++) printf("%s%g", i ? ", " : "", o7[i]);
    printf("],\n \"world_in\": [");
    for (int i = 0; i < 12; i++) printf("%s%g", i ? ", " : "", p12[i]);
    printf("],\n \"world_t\": [%g, %g, %g],\n \"world_R_rowmajor\": [", t[0], t[1], t[2]);
    for (int i = 0; i < 9; i++) printf("%s%g", i ? ", " : "", R[i]);
    printf("],\n \"world_inv_t\": [%g, %g, %g],\n \"world_inv_R_rowmajor\": [", ti[0], ti[1], ti[2]);
    for (int i = 0; i < 9; i++) printf("%s%g", i ? ", " : "", Ri[i]);
    printf("],\n \"icp_default_maxIter\": 10,\n \"prototypes_checked\": [\"generateVerticesFromDepthMap\", \"generateMeshFromDepthMaps\", "
           "\"depthMapAndColorSetRadialCorrection\", \"deleteMesh\", \"ICP\"]\n}\n");
    return 0;
}
#endif
