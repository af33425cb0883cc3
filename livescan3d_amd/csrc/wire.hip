// wire.hip -- the data formats either side of the fusion path, part 3 of include/NativeUtils.h.
//
//   inbound  : the client's frame message (LiveScanClient::SerializeFrame, src/LiveScanClient/liveScanClient.cpp:185-290,
//              read by KinectSocket.ReceiveFrame, LiveScanServer/KinectSocket.cs:211-304) and the client's recording file
//              (src/LiveScanClient/frameFileWriterReader.cpp:59-82,115-130).  Host-side parsing only; zstd is the system
//              libzstd.so.1 loaded at run time (the reference P/Invokes libzstd.dll the same way, ZSTDDecompressor.cs:13-31).
//   outbound : the TransferServer stream (formVerticesChunks / formMeshChunks, LiveScanServer/TransferServer.cs:177-270, and
//              TransferSocket.SendFrame, LiveScanServer/TransferSocket.cs:50-104) and the binary PLY writer
//              (LiveScanServer/Utils.cs:222-262), built ON THE DEVICE from a cloud / mesh that is already in HBM, so that one
//              D2H copy yields the bytes to put on the socket / in the file.
//
// formMeshChunks is a sequential re-indexing loop in the reference (a vertex is re-emitted the first time a chunk uses it,
// a chunk closes at the first triangle end with >= 64997 vertices).  Here every index position of a window is processed in
// parallel: positions claim their vertex with a 64-bit atomicMin of (chunk, position) -- the winner is the first use inside
// the chunk --, a block-sum + last-block scan finds the triangle that closes the chunk, and the new indices are the ranks of
// the winners.  Only the chunk boundaries are found one after the other (the boundary of chunk c+1 depends on chunk c).
#include "lsn_common.hpp"

#include <dlfcn.h>

#include <mutex>
#include <vector>

namespace {

constexpr int kChunkLimit = 65000 - 3;                       // TransferServer.cs:179,205
constexpr int kTriPerBlock = 256;                            // one triangle (3 index positions) per thread
constexpr int kWindowBlocks = 768;
constexpr int kWindowTri = kWindowBlocks * kTriPerBlock;     // triangles examined per iteration (a chunk of a grid mesh spans ~130 k)
constexpr int kItemsPerBlock = 1024;                         // assemble kernels: vertices / faces per block (4 per thread)

struct ChunkState {
    int s_tri;       // first triangle of the current chunk
    int win_tri;     // first triangle of the current window (>= s_tri)
    int c;           // current chunk id
    int vbase;       // vertices emitted by closed chunks
    int carry;       // vertices of the current chunk found by earlier windows
    int tri_start;   // the reference's trianglesChunkStart (index-position units, :213,:244)
    int done;
    int bad;         // an index outside [0, nV) was seen
    int e_tri;       // count pass: last triangle of this window that belongs to the current chunk
    int closes;      //             the chunk ends at e_tri
    int vcount;      //             vertices of the chunk through e_tri (carry included)
    int p_first;     // triangles [p_first, p_last] wait for their new indices (written by the next tag pass), -1 none
    int p_last;
    unsigned arrive_count;
    unsigned arrive_emit;
    int pad;
};

__device__ __forceinline__ unsigned long long chunk_key(int c, int pos)
{
    return ((unsigned long long)(0xFFFFFFu - (unsigned)c) << 32) | (unsigned)pos;
}

// ---- pass 1: (a) new indices of the previous window, (b) claim the vertices of this window -----------------------------
__global__ __launch_bounds__(kTriPerBlock) void chunk_tag_kernel(ChunkState *st, const int *__restrict__ tri, int nT, int nV,
                                                                   unsigned long long *tag, const int *__restrict__ lidx,
                                                                   int *__restrict__ new_tri)
{
    const int tid = blockIdx.x * kTriPerBlock + threadIdx.x;
    const int pf = st->p_first, pl = st->p_last;
    if (pf >= 0 && pf + tid <= pl) {
        const int k = pf + tid;
#pragma unroll
        for (int j = 0; j < 3; j++) {
            const int val = tri[3 * k + j];
            new_tri[3 * k + j] = (unsigned)val < (unsigned)nV ? lidx[val] : -1;
        }
    }
    if (st->done) return;
    const int k = st->win_tri + tid;
    if (k >= nT) return;
    const int c = st->c;
    bool bad = false;
#pragma unroll
    for (int j = 0; j < 3; j++) {
        const int val = tri[3 * k + j];
        if ((unsigned)val < (unsigned)nV)
            atomicMin(&tag[val], chunk_key(c, 3 * k + j));
        else
            bad = true;
    }
    if (bad) atomicOr(&st->bad, 1);
}

__device__ __forceinline__ int triangle_new_mask(const int *__restrict__ tri, int k, int nV, int c, const unsigned long long *tag)
{
    int m = 0;
#pragma unroll
    for (int j = 0; j < 3; j++) {
        const int val = tri[3 * k + j];
        if ((unsigned)val < (unsigned)nV && tag[val] == chunk_key(c, 3 * k + j)) m |= 1 << j;
    }
    return m;
}

// What the workgroups of ONE launch tell each other here goes through agent-scope (write-through) atomic stores and loads, and a
// workgroup counts itself in once those stores have been acknowledged: vmcnt(0).  Not __threadfence(): an agent-scope fence writes back
// and invalidates the XCD's whole L2 (buffer_wbl2 / buffer_inv), per wave that executes it -- measured on ICP's match step, where two
// such fences per workgroup made one fused launch cost 44 us against 24 us for the three launches it replaced
// (profiles/r05_ab_icp_fused_match.txt).  Everything else these kernels write is read by LATER launches.
// This rests on two gfx9 (gfx942 / gfx950) facts, not on the HSA memory model: agent-scope atomics are sc1 write-through accesses that
// reach the memory side, and s_waitcnt's vmcnt counts STORES as well as loads -- on gfx10+ stores have their own counter (vscnt) and
// s_waitcnt(0) would not wait for them.  The library is built for gfx950 only (csrc/Makefile); any other target stops here.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__) && !defined(__gfx942__)
#error "stores_acknowledged() relies on gfx9 semantics (vmcnt covers stores, sc1 write-through atomics): build for gfx950"
#endif
__device__ __forceinline__ void stores_acknowledged()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_s_waitcnt(0);
}

// inclusive scan over the block's 256 threads (4 waves)
__device__ __forceinline__ int block_inclusive_scan(int v, int *s_wave /* 4 ints */)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int x = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int y = __shfl_up(x, d, 64);
        if (lane >= d) x += y;
    }
    if (lane == 63) s_wave[wave] = x;
    __syncthreads();
    int add = 0;
    for (int w2 = 0; w2 < wave; w2++) add += s_wave[w2];
    __syncthreads();
    return x + add;
}

// ---- pass 2: vertices first used per block; the last block to arrive finds where the chunk closes --------------------------
__global__ __launch_bounds__(kTriPerBlock) void chunk_count_kernel(ChunkState *st, const int *__restrict__ tri, int nT, int nV,
                                                                     const unsigned long long *tag, int *bsum, int *boff)
{
    __shared__ int s_wave[4];
    __shared__ int s_flag;
    __shared__ int s_found_block, s_found_off;
    if (st->done) return;
    const int win = st->win_tri, c = st->c;
    const int k = win + blockIdx.x * kTriPerBlock + threadIdx.x;
    const int cnt = k < nT ? __popc(triangle_new_mask(tri, k, nV, c, tag)) : 0;
    const int incl = block_inclusive_scan(cnt, s_wave);
    if (threadIdx.x == kTriPerBlock - 1) {
        __hip_atomic_store(&bsum[blockIdx.x], incl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        stores_acknowledged();
        s_flag = atomicAdd(&st->arrive_count, 1u) == gridDim.x - 1;
    }
    __syncthreads();
    if (!s_flag) return;
    // ---- last block: exclusive offsets of the blocks, and the block in which the running count reaches the limit
    const int carry = st->carry;
    const int nb = gridDim.x;
    if (threadIdx.x == 0) { s_found_block = -1; s_found_off = 0; }
    __syncthreads();
    int running = carry;
    for (int b0 = 0; b0 < nb; b0 += kTriPerBlock) {
        const int b = b0 + threadIdx.x;
        const int v = b < nb ? __hip_atomic_load(&bsum[b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;
        const int inc = block_inclusive_scan(v, s_wave) + running;
        if (b < nb) {
            boff[b] = inc - v;
            if (inc >= kChunkLimit && inc - v < kChunkLimit) { s_found_block = b; s_found_off = inc - v; }   // unique: inc is monotone
        }
        if (threadIdx.x == kTriPerBlock - 1) s_flag = inc;
        __syncthreads();
        running = s_flag;
        __syncthreads();
    }
    const int fb = s_found_block;
    if (fb >= 0) {
        // the triangle inside block fb at which the count reaches the limit: the chunk closes at its last index (:239)
        const int kk = win + fb * kTriPerBlock + threadIdx.x;
        const int c2 = kk < nT ? __popc(triangle_new_mask(tri, kk, nV, c, tag)) : 0;
        const int inc2 = block_inclusive_scan(c2, s_wave) + s_found_off;
        if (inc2 >= kChunkLimit && inc2 - c2 < kChunkLimit) {
            st->e_tri = kk;
            st->closes = 1;
            st->vcount = inc2;
        }
    } else if (threadIdx.x == 0) {
        const int nwin = min(kWindowTri, nT - win);
        st->e_tri = win + nwin - 1;
        st->closes = 0;
        st->vcount = running;
    }
    if (threadIdx.x == 0) st->arrive_count = 0;
}

// ---- pass 3: emit the chunk's vertices of this window; the last block to arrive advances the state ------------------------
__global__ __launch_bounds__(kTriPerBlock) void chunk_emit_kernel(ChunkState *st, const int *__restrict__ tri, int nT, int nV,
                                                                    const uint4 *__restrict__ verts, const unsigned long long *tag,
                                                                    const int *__restrict__ boff, int *lidx, uint4 *__restrict__ new_v,
                                                                    int *v_chunks, int *t_chunks)
{
    __shared__ int s_wave[4];
    __shared__ int s_flag;
    if (st->done) return;
    const int win = st->win_tri, c = st->c, e = st->e_tri, vbase = st->vbase;
    const int k = win + blockIdx.x * kTriPerBlock + threadIdx.x;
    const int m = (k <= e && k < nT) ? triangle_new_mask(tri, k, nV, c, tag) : 0;
    const int cnt = __popc(m);
    int li = block_inclusive_scan(cnt, s_wave) - cnt + boff[blockIdx.x];
#pragma unroll
    for (int j = 0; j < 3; j++) {
        if (m & (1 << j)) {
            const int val = tri[3 * k + j];
            new_v[vbase + li] = verts[val];                   // newVertices[currentVertex] = lVertices[val]  (:234)
            lidx[val] = li;                                   // verticesMap[val] = verticesInCurrentChunk     (:235)
            li++;
        }
    }
    if (threadIdx.x == 0) s_flag = atomicAdd(&st->arrive_emit, 1u) == gridDim.x - 1;
    __syncthreads();
    if (!s_flag || threadIdx.x != 0) return;
    st->arrive_emit = 0;
    st->p_first = win;
    st->p_last = e;
    const int vcount = st->vcount;
    if (st->closes) {
        const int t = 3 * e + 2;                              // the t of :239-246
        v_chunks[c] = vcount;
        t_chunks[c] = (t - st->tri_start) / 3;
        st->tri_start = t;                                    // sic (:244)
        st->vbase = vbase + vcount;
        st->c = c + 1;
        st->carry = 0;
        st->s_tri = e + 1;
        st->win_tri = e + 1;
        if (e + 1 >= nT) st->done = 1;
    } else {
        st->carry = vcount;
        st->win_tri = e + 1;
        if (e + 1 >= nT) {                                    // :256-260
            if (vcount != 0) {
                v_chunks[c] = vcount;
                t_chunks[c] = (3 * nT - st->tri_start) / 3;
                st->c = c + 1;
                st->vbase = vbase + vcount;
            }
            st->done = 1;
        }
    }
}

// ---- the link path: formMeshChunks without a chunk-by-chunk walk --------------------------------------------------------------------
//
// The window walk above finds the chunks one after the other: three dependent launches per chunk, ~50 for the mesh of a tick.  For meshes
// whose vertex re-use is LOCAL -- every vertex used at most 16 times, two consecutive uses less than a chunk's worth of index positions
// apart: every grid mesh -- the chunks follow from one prefix sum:
//   * prev[pos] = the previous index position that uses the same vertex (-1: none), last[pos] = no later position does.  A position opens
//     a new vertex in the chunk that starts at triangle s iff prev[pos] < 3 s;
//   * PF[t] = vertices used for the first time up to and including triangle t, PH[t] = vertices used for the last time up to and
//     including t.  The chunk that starts at triangle s holds, through triangle e, PF[e] - PH[s - 1] vertices: those that have begun minus
//     those that had already ended -- exact unless a vertex is used before s and after e but not in between, which a link shorter than
//     64997 positions rules out from e = s + 21665 on, and a chunk cannot close earlier (a triangle opens at most three vertices).  The end
//     of a chunk is therefore a search in PF; the chunks of a mesh are ~13 such searches by one wave;
//   * the new index of a position = its rank among the chunk's opening positions (one more prefix sum), or the rank of the opening
//     position its prev-chain leads to.
// A vertex with more than 16 uses or a longer link sends the call down the window walk instead (same bytes).
constexpr int kMaxUses = 16;
constexpr int kScanPerBlock = 2048;   // elements per workgroup of the prefix sums (8 per thread)
constexpr unsigned kLastUse = 0x80000000u;   // prev[] holds (previous position + 1) | kLastUse

struct LinkState {
    int n_chunks, n_send, bad, fallback;
};

__device__ __forceinline__ int link_prev_of(unsigned code) { return (int)(code & ~kLastUse) - 1; }

// uses[v][k] = the index positions that use vertex v, in no particular order.  A workgroup takes 12288 consecutive positions (4096
// triangles, a few rows of a grid mesh), counts the vertices within 8192 ids of its smallest one in LDS and reserves each vertex's slots in
// the global row with ONE atomic per vertex it touches; ids outside the window take one atomic per position.  (Device-scope atomics execute
// at the memory side: one returning atomic per position costs 56 us for a tick's mesh, this 23 us, of which the scattered stores are most.)
constexpr int kUsesThreads = 1024, kUsesPer = 12, kUsesWindow = 8192;

__global__ __launch_bounds__(kUsesThreads) void link_uses_kernel(const int *__restrict__ tri, int n_pos, int nV, int *cnt, int *uses, LinkState *st)
{
    __shared__ int s_cnt[kUsesWindow];
    __shared__ int s_base[kUsesWindow];
    __shared__ int s_min[kUsesThreads / 64];
    const int tid = threadIdx.x;
    const long long block0 = (long long)blockIdx.x * (kUsesThreads * kUsesPer);
    int val[kUsesPer], kl[kUsesPer];
    int vmin = 0x7fffffff;
    bool bad = false;
#pragma unroll
    for (int i = 0; i < kUsesPer; i++) {
        const long long pos = block0 + i * kUsesThreads + tid;
        val[i] = pos < n_pos ? tri[pos] : -1;
        if (pos < n_pos && (unsigned)val[i] >= (unsigned)nV) { bad = true; val[i] = -1; }
        if (val[i] >= 0) vmin = min(vmin, val[i]);
    }
    if (bad) atomicOr(&st->bad, 1);
    for (int i = tid; i < kUsesWindow; i += kUsesThreads) s_cnt[i] = 0;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) vmin = min(vmin, __shfl_xor(vmin, d, 64));
    if ((tid & 63) == 0) s_min[tid >> 6] = vmin;
    __syncthreads();
    vmin = s_min[0];
#pragma unroll
    for (int w2 = 1; w2 < kUsesThreads / 64; w2++) vmin = min(vmin, s_min[w2]);
#pragma unroll
    for (int i = 0; i < kUsesPer; i++) {
        const unsigned d = (unsigned)val[i] - (unsigned)vmin;
        kl[i] = val[i] >= 0 && d < (unsigned)kUsesWindow ? atomicAdd(&s_cnt[d], 1) : -1;
    }
    __syncthreads();
    for (int d = tid; d < kUsesWindow; d += kUsesThreads) {
        const int n = s_cnt[d];
        if (n) s_base[d] = atomicAdd(&cnt[vmin + d], n);
    }
    __syncthreads();
    bool over = false;
#pragma unroll
    for (int i = 0; i < kUsesPer; i++) {
        if (val[i] < 0) continue;
        const long long pos = block0 + i * kUsesThreads + tid;
        const int k = kl[i] >= 0 ? s_base[val[i] - vmin] + kl[i] : atomicAdd(&cnt[val[i]], 1);
        if (k < kMaxUses) uses[(size_t)val[i] * kMaxUses + k] = (int)pos;
        else over = true;
    }
    if (over) atomicOr(&st->fallback, 1);
}

// the uses of one vertex, each against all: the previous use = the largest smaller one (no sort), the last use = none larger
template <int N>
__device__ __forceinline__ void link_row(const int (&u)[16], int n, unsigned *prev, LinkState *st)
{
#pragma unroll
    for (int i = 0; i < N; i++) {
        if (i < n) {
            const int p = u[i];
            int best = -1;
            bool later = false;
#pragma unroll
            for (int j = 0; j < N; j++) {
                const int q = u[j];                               // -2 where the row has no entry
                best = (q < p && q > best) ? q : best;
                later |= q > p;
            }
            prev[p] = (unsigned)(best + 1) | (later ? 0u : kLastUse);
            if (best >= 0 && p - best >= kChunkLimit) atomicOr(&st->fallback, 1);   // a link as long as a chunk: the closed form does not hold
        }
    }
}

__global__ __launch_bounds__(256) void link_prev_kernel(const int *__restrict__ cnt, const int *__restrict__ uses, int nV, unsigned *prev, LinkState *st)
{
    const int v = blockIdx.x * 256 + threadIdx.x;
    const int n = v < nV ? min(cnt[v], kMaxUses) : 0;
    const int4 *row = reinterpret_cast<const int4 *>(uses + (size_t)min(v, nV - 1) * kMaxUses);
    int u[16];
    const bool wide = __any(n > 8);
    {
        const int4 a = row[0], b = row[1];
        u[0] = a.x; u[1] = a.y; u[2] = a.z; u[3] = a.w; u[4] = b.x; u[5] = b.y; u[6] = b.z; u[7] = b.w;
    }
    if (wide) {
        const int4 a = row[2], b = row[3];
        u[8] = a.x; u[9] = a.y; u[10] = a.z; u[11] = a.w; u[12] = b.x; u[13] = b.y; u[14] = b.z; u[15] = b.w;
    }
#pragma unroll
    for (int i = 0; i < 16; i++)
        if (i >= n) u[i] = -2;
    if (wide) link_row<16>(u, n, prev, st);
    else link_row<8>(u, n, prev, st);
}

// one thread per triangle: how many of its positions are first uses (low half) and last uses (high half)
__global__ __launch_bounds__(256) void link_pack_kernel(const unsigned *__restrict__ prev, int nT, unsigned long long *pack, const LinkState *st)
{
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= nT || (st->bad | st->fallback)) return;                  // prev[] is incomplete then: nothing below may follow it
    unsigned f = 0, h = 0;
#pragma unroll
    for (int j = 0; j < 3; j++) {
        const unsigned c = prev[3 * t + j];
        f += (c & ~kLastUse) == 0;
        h += c >> 31;
    }
    pack[t] = (unsigned long long)f | ((unsigned long long)h << 32);
}

// ---- prefix sums over a few million elements: block totals, one workgroup over the totals, blocks again ----
template <typename T>
__device__ __forceinline__ T block_scan_incl(T v, T *s_wave /* 4 */)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    T x = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const T y = __shfl_up(x, d, 64);
        if (lane >= d) x += y;
    }
    if (lane == 63) s_wave[wave] = x;
    __syncthreads();
    T add = 0;
    for (int w2 = 0; w2 < wave; w2++) add += s_wave[w2];
    __syncthreads();
    return x + add;
}

// eight consecutive elements of a thread (the arrays are 32-byte aligned and padded to a multiple of eight)
template <typename T>
__device__ __forceinline__ void load8(const T *in, long long base, long long n, T (&v)[8])
{
    static_assert(sizeof(T) == 4 || sizeof(T) == 8, "");
    constexpr int kVec = 16 / sizeof(T);
    const uint4 *src = reinterpret_cast<const uint4 *>(in + base);
    uint4 raw[8 / kVec];
    if (base < n) {
#pragma unroll
        for (int i = 0; i < 8 / kVec; i++) raw[i] = src[i];
    }
    const T *r = reinterpret_cast<const T *>(raw);
#pragma unroll
    for (int k = 0; k < 8; k++) v[k] = base + k < n ? r[k] : T(0);
}

template <typename T>
__device__ __forceinline__ void store8(T *out, long long base, long long n, const T (&v)[8])
{
    constexpr int kVec = 16 / sizeof(T);
    if (base + 8 <= n) {
        const uint4 *r = reinterpret_cast<const uint4 *>(v);
        uint4 *dst = reinterpret_cast<uint4 *>(out + base);
#pragma unroll
        for (int i = 0; i < 8 / kVec; i++) dst[i] = r[i];
    } else {
#pragma unroll
        for (int k = 0; k < 8; k++)
            if (base + k < n) out[base + k] = v[k];
    }
}

template <typename T>
__global__ __launch_bounds__(256) void scan_totals_kernel(const T *__restrict__ in, long long n, T *totals)
{
    __shared__ T s_wave[4];
    const long long base = (long long)blockIdx.x * kScanPerBlock + threadIdx.x * 8;
    T v[8], sum = 0;
    load8(in, base, n, v);
#pragma unroll
    for (int k = 0; k < 8; k++) sum += v[k];
    const T incl = block_scan_incl(sum, s_wave);
    if (threadIdx.x == 255) totals[blockIdx.x] = incl;
}

template <typename T>
__global__ __launch_bounds__(1024) void scan_top_kernel(T *totals, int nb)   // totals -> exclusive prefixes, in place
{
    __shared__ T s_wave[16];
    __shared__ T s_carry;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) s_carry = 0;
    __syncthreads();
    for (int b0 = 0; b0 < nb; b0 += 1024) {
        const int b = b0 + threadIdx.x;
        const T v = b < nb ? totals[b] : T(0);
        T x = v;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const T y = __shfl_up(x, d, 64);
            if (lane >= d) x += y;
        }
        if (lane == 63) s_wave[wave] = x;
        __syncthreads();
        T add = s_carry;
        for (int w2 = 0; w2 < wave; w2++) add += s_wave[w2];
        if (b < nb) totals[b] = x + add - v;
        __syncthreads();
        if (threadIdx.x == 1023) s_carry = x + add;
        __syncthreads();
    }
}

template <typename T>
__global__ __launch_bounds__(256) void scan_apply_kernel(T *data /* in place, inclusive */, long long n, const T *__restrict__ totals)
{
    __shared__ T s_wave[4];
    const long long base = (long long)blockIdx.x * kScanPerBlock + threadIdx.x * 8;
    T v[8], sum = 0;
    load8(data, base, n, v);
#pragma unroll
    for (int k = 0; k < 8; k++) sum += v[k];
    T run = block_scan_incl(sum, s_wave) - sum + totals[blockIdx.x];
#pragma unroll
    for (int k = 0; k < 8; k++) {
        run += v[k];
        v[k] = run;
    }
    store8(data, base, n, v);
}

// the first index in [lo, hi] whose key reaches `need` (keys non-decreasing, key(get(hi)) >= need), by the whole wave; *found = get(index)
template <class V, class Get, class Key>
__device__ __forceinline__ int wave_first(int lo, int hi, int need, Get get, Key key, V *found)
{
    const int lane = threadIdx.x & 63;
    while (hi - lo >= 64) {
        const int step = (hi - lo + 64) / 65;                          // probes lo + step, lo + 2 step, ... clipped to hi
        const int probe = min(lo + (lane + 1) * step, hi);             // (no overflow: callers search ranges far below 2^31 / 64)
        const unsigned long long m = __ballot(key(get(probe)) >= need);
        if (m == 0) {
            lo = lo + 64 * step + 1;
        } else {
            const int first = __ffsll((long long)m) - 1;
            hi = min(lo + (first + 1) * step, hi);
            lo = first == 0 ? lo : lo + first * step + 1;
        }
    }
    const V v = get(min(lo + lane, hi));
    const unsigned long long m = __ballot(key(v) >= need);
    const int first = __ffsll((long long)m) - 1;
    *found = __shfl(v, first, 64);
    return lo + first;
}

constexpr int kLinkChunksLds = 1024;      // chunk starts a workgroup keeps in LDS
constexpr int kLinkTotalsLds = 12288;     // block totals the chunk search keeps in LDS (>= 1024 chunks' worth of triangles / 2048)

// one wave: the chunk ends, one search per chunk (TransferServer.cs:239-260 with its counters in closed form)
__global__ __launch_bounds__(64) void link_chunks_kernel(const unsigned long long *__restrict__ P /* inclusive prefix of pack[], nT entries */,
                                                         const unsigned long long *__restrict__ totals /* exclusive, per 2048 triangles */, int nT,
                                                         int max_chunks, int *chunk_start /* [max_chunks + 1] */, int *chunk_vbase, int *v_chunks,
                                                         int *t_chunks, LinkState *st)
{
    __shared__ int s_tot[kLinkTotalsLds];
    const int lane = threadIdx.x;
    if (st->bad | st->fallback) return;
    const int nb = (nT + kScanPerBlock - 1) / kScanPerBlock;
    for (int b = lane; b < nb; b += 64) s_tot[b] = (int)(unsigned)totals[b];                 // PF before block b
    __syncthreads();
    auto low = [](unsigned long long v) { return (int)(unsigned)v; };
    int s = 0, c = 0, vbase = 0, tstart = 0;
    int ended = 0;                                                     // vertices whose last use lies before the chunk: PH[s - 1]
    const int pf_last = low(P[nT - 1]);
    while (s < nT && c < max_chunks) {
        const int need = kChunkLimit + ended;                          // the chunk closes at the first e with PF[e] >= need ...
        const int lo = s + (kChunkLimit + 2) / 3 - 1;                  // ... which is never before this triangle
        if (lo > nT - 1 || pf_last < need) {                           // the mesh ends first (:256-260)
            const int vcount = pf_last - ended;
            if (lane == 0) {
                chunk_start[c] = s;
                chunk_vbase[c] = vbase;
                if (vcount != 0) {
                    v_chunks[c] = vcount;
                    t_chunks[c] = (3 * nT - tstart) / 3;
                }
            }
            if (vcount != 0) { c++; vbase += vcount; }
            s = nT;
            break;
        }
        // the block of 2048 triangles in which PF reaches `need`: the last one whose prefix is still short of it
        int b = nb - 1, unused;
        if (s_tot[nb - 1] >= need)                                     // s_tot[0] = 0 < need: the first block that is not short is >= 1
            b = wave_first(0, nb - 1, need, [&](int i) { return s_tot[i]; }, [](int v) { return v; }, &unused) - 1;
        unsigned long long at_e;
        int e = wave_first(b * kScanPerBlock, min(b * kScanPerBlock + kScanPerBlock - 1, nT - 1), need, [&](int t) { return P[t]; }, low, &at_e);
        if (e < lo) {                                                  // the closed form over-counts before `lo` (vertices used on both sides of a short window only); it is monotone, so then the chunk closes at `lo`
            e = lo;
            at_e = P[e];
        }
        const int vcount = low(at_e) - ended;
        const int t = 3 * e + 2;
        if (lane == 0) {
            chunk_start[c] = s;
            chunk_vbase[c] = vbase;
            v_chunks[c] = vcount;
            t_chunks[c] = (t - tstart) / 3;
        }
        tstart = t;                                                    // sic (:244)
        vbase += vcount;
        ended = (int)(at_e >> 32);                                     // PH[e]: what has ended before the next chunk
        c++;
        s = e + 1;
    }
    if (lane == 0) {
        chunk_start[c] = nT;
        if (s < nT) st->fallback = 1;                                  // more chunks than the tables hold
        st->n_chunks = c;
        st->n_send = vbase;
    }
}

__device__ __forceinline__ int chunk_of(const int *s_start, int n_chunks, int t)
{
    int lo = 0, hi = n_chunks - 1;                                     // last chunk with start <= t
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (s_start[mid] <= t) lo = mid;
        else hi = mid - 1;
    }
    return lo;
}

// which of a thread's eight positions open a vertex in their chunk (s_start[n_chunks] = nT)
__device__ __forceinline__ void opening_flags(const unsigned *prev, long long base, int n_pos, const int *s_start, int nc, int (&flag)[8])
{
    unsigned code[8];
    load8(prev, base, (long long)n_pos, code);
    int c = base < n_pos ? chunk_of(s_start, nc, (int)(base / 3)) : 0;
#pragma unroll
    for (int k = 0; k < 8; k++) {
        const int pos = (int)base + k;
        if (pos < n_pos && pos / 3 >= s_start[c + 1]) c++;
        flag[k] = pos < n_pos && link_prev_of(code[k]) < 3 * s_start[c];
    }
}

__device__ __forceinline__ bool load_chunk_starts(int *s_start, const int *chunk_start, const LinkState *st, int &nc)
{
    if (st->bad | st->fallback) return false;
    nc = st->n_chunks;
    for (int i = threadIdx.x; i <= nc && i <= kLinkChunksLds; i += 256) s_start[i] = chunk_start[i];
    __syncthreads();
    return true;
}

__global__ __launch_bounds__(256) void rank_totals_kernel(const unsigned *__restrict__ prev, int n_pos, const int *__restrict__ chunk_start,
                                                          const LinkState *st, int *totals)
{
    __shared__ int s_start[kLinkChunksLds + 1];
    __shared__ int s_wave[4];
    int nc;
    if (!load_chunk_starts(s_start, chunk_start, st, nc)) return;
    const long long base = (long long)blockIdx.x * kScanPerBlock + threadIdx.x * 8;
    int flag[8], sum = 0;
    opening_flags(prev, base, n_pos, s_start, nc, flag);
#pragma unroll
    for (int k = 0; k < 8; k++) sum += flag[k];
    const int incl = block_scan_incl(sum, s_wave);
    if (threadIdx.x == 255) totals[blockIdx.x] = incl;
}

// rank[pos] = opening positions before pos (exclusive prefix over the whole mesh)
__global__ __launch_bounds__(256) void rank_apply_kernel(const unsigned *__restrict__ prev, int n_pos, const int *__restrict__ chunk_start,
                                                         const LinkState *st, const int *__restrict__ totals, int *__restrict__ rank)
{
    __shared__ int s_start[kLinkChunksLds + 1];
    __shared__ int s_wave[4];
    int nc;
    if (!load_chunk_starts(s_start, chunk_start, st, nc)) return;
    const long long base = (long long)blockIdx.x * kScanPerBlock + threadIdx.x * 8;
    int flag[8], sum = 0;
    opening_flags(prev, base, n_pos, s_start, nc, flag);
#pragma unroll
    for (int k = 0; k < 8; k++) sum += flag[k];
    int run = block_scan_incl(sum, s_wave) - sum + totals[blockIdx.x];
#pragma unroll
    for (int k = 0; k < 8; k++) {
        const int f = flag[k];
        flag[k] = run;
        run += f;
    }
    store8(rank, base, (long long)n_pos, flag);
}

__global__ __launch_bounds__(256) void link_emit_kernel(const int *__restrict__ tri, const unsigned *__restrict__ prev, int n_pos,
                                                        const uint4 *__restrict__ verts, const int *__restrict__ chunk_start,
                                                        const int *__restrict__ chunk_vbase, const LinkState *st,
                                                        const int *__restrict__ rank, uint4 *__restrict__ new_v, int *__restrict__ new_tri)
{
    __shared__ int s_start[kLinkChunksLds + 1];
    int nc;
    if (!load_chunk_starts(s_start, chunk_start, st, nc)) return;
    const int pos = blockIdx.x * 256 + threadIdx.x;
    if (pos >= n_pos) return;
    const int c = chunk_of(s_start, nc, pos / 3);
    const int p0 = 3 * s_start[c];
    const int r0 = rank[p0];
    int q = pos;
    for (int hop = 0; hop < kMaxUses; hop++) {                              // the position that opened this vertex in the chunk
        const int before = link_prev_of(prev[q]);
        if (before < p0) break;
        q = before;
    }
    const int li = rank[q] - r0;
    new_tri[pos] = li;                                                       // verticesMap[val] / verticesInCurrentChunk (:231-241)
    if (q == pos) new_v[chunk_vbase[c] + li] = verts[tri[pos]];              // newVertices[currentVertex] = lVertices[val] (:234)
}

// ---- byte-stream stores -------------------------------------------------------------------------------------------------------
// Writes nbytes composed in LDS (dword 0 = stream byte 0; one readable dword past the end) to an arbitrarily aligned
// destination: single bytes up to the first aligned dword, then aligned dwords rebuilt from two LDS words with
// v_alignbyte_b32, then the tail bytes.  Coalesced regardless of the section's byte offset inside the message.
__device__ __forceinline__ unsigned lds_byte(const unsigned *lds, int i) { return (lds[i >> 2] >> (8 * (i & 3))) & 0xffu; }

__device__ __forceinline__ void block_stream_store(unsigned char *dst, const unsigned *lds, int nbytes)
{
    int head = (int)((4 - ((size_t)dst & 3)) & 3);
    if (head > nbytes) head = nbytes;
    if ((int)threadIdx.x < head) dst[threadIdx.x] = (unsigned char)lds_byte(lds, threadIdx.x);
    const int nw = (nbytes - head) >> 2;
    unsigned *d32 = reinterpret_cast<unsigned *>(dst + head);
    for (int j = threadIdx.x; j < nw; j += blockDim.x) {
        const unsigned lo = lds[j], hi = lds[j + 1];
        d32[j] = head ? __builtin_amdgcn_alignbyte(hi, lo, (unsigned)head) : lo;
    }
    for (int i = head + 4 * nw + threadIdx.x; i < nbytes; i += blockDim.x) dst[i] = (unsigned char)lds_byte(lds, i);
}

// ORs a 32-bit value into a word array at a compile-time byte offset
template <int OFF> __device__ __forceinline__ void put32(unsigned *w, unsigned v)
{
    w[OFF >> 2] |= v << (8 * (OFF & 3));
    if constexpr ((OFF & 3) != 0) w[(OFF >> 2) + 1] |= v >> (32 - 8 * (OFF & 3));
}
template <int OFF> __device__ __forceinline__ void put8(unsigned *w, unsigned v) { w[OFF >> 2] |= (v & 0xffu) << (8 * (OFF & 3)); }

template <int I> __device__ __forceinline__ void ply_vertex(unsigned *w, const uint4 v)
{
    // {R,G,B,A | X | Y | Z} -> {X, Y, Z, R, G, B} = 15 bytes (Utils.cs:248-254)
    put32<15 * I + 0>(w, v.y);
    put32<15 * I + 4>(w, v.z);
    put32<15 * I + 8>(w, v.w);
    put8<15 * I + 12>(w, v.x);
    put8<15 * I + 13>(w, v.x >> 8);
    put8<15 * I + 14>(w, v.x >> 16);
}

template <int I> __device__ __forceinline__ void ply_face(unsigned *w, int a, int b, int c)
{
    put8<13 * I + 0>(w, 3u);                                  // (byte)3, then the three indices (Utils.cs:259-262)
    put32<13 * I + 1>(w, (unsigned)a);
    put32<13 * I + 5>(w, (unsigned)b);
    put32<13 * I + 9>(w, (unsigned)c);
}

struct PlyHeader {
    int len;
    char text[316];
};

// blocks [0, vb) : vertex records, [vb, vb+fb) : face records, last block: the header text
__global__ __launch_bounds__(256) void ply_pack_kernel(const uint4 *__restrict__ verts, int nV, const int *__restrict__ tri, int nT,
                                                        unsigned char *__restrict__ out, PlyHeader hdr, int vb, int fb)
{
    __shared__ unsigned lds[kItemsPerBlock * 15 / 4 + 1];
    const int b = blockIdx.x;
    if (b < vb) {
        const int first = b * kItemsPerBlock, n = min(kItemsPerBlock, nV - first);
        const int i0 = first + 4 * threadIdx.x;
        unsigned w[16];
#pragma unroll
        for (int q = 0; q < 16; q++) w[q] = 0;
        const uint4 z = make_uint4(0, 0, 0, 0);
        ply_vertex<0>(w, i0 + 0 < nV ? verts[i0 + 0] : z);
        ply_vertex<1>(w, i0 + 1 < nV ? verts[i0 + 1] : z);
        ply_vertex<2>(w, i0 + 2 < nV ? verts[i0 + 2] : z);
        ply_vertex<3>(w, i0 + 3 < nV ? verts[i0 + 3] : z);
#pragma unroll
        for (int q = 0; q < 15; q++) lds[15 * threadIdx.x + q] = w[q];
        __syncthreads();
        block_stream_store(out + hdr.len + 15ll * first, lds, 15 * n);
    } else if (b < vb + fb) {
        const int first = (b - vb) * kItemsPerBlock, n = min(kItemsPerBlock, nT - first);
        const int i0 = first + 4 * threadIdx.x;
        unsigned w[14];
#pragma unroll
        for (int q = 0; q < 14; q++) w[q] = 0;
        int t[12];
#pragma unroll
        for (int q = 0; q < 12; q++) t[q] = (i0 + q / 3) < nT ? tri[3 * i0 + q] : 0;
        ply_face<0>(w, t[0], t[1], t[2]);
        ply_face<1>(w, t[3], t[4], t[5]);
        ply_face<2>(w, t[6], t[7], t[8]);
        ply_face<3>(w, t[9], t[10], t[11]);
#pragma unroll
        for (int q = 0; q < 13; q++) lds[13 * threadIdx.x + q] = w[q];
        __syncthreads();
        block_stream_store(out + hdr.len + 15ll * nV + 13ll * first, lds, 13 * n);
    } else {
        for (int i = threadIdx.x; i < hdr.len; i += blockDim.x) out[i] = (unsigned char)hdr.text[i];
    }
}

// SendFrame stream: [3 ints][v chunk sizes][t chunk sizes][xyz f32 x 3 x n][rgb u8 x 3 x n][tri i32 x 3 x nT]
// blocks [0, xb): xyz, [xb, xb+cb): rgb, [.., +tb): triangles (1024 ints per block... 3072 bytes), last block: the head
__global__ __launch_bounds__(256) void transfer_assemble_kernel(const uint4 *__restrict__ verts, int n_send, const int *__restrict__ tri,
                                                                 int n_tri, int n_chunks, const int *__restrict__ v_chunks,
                                                                 const int *__restrict__ t_chunks, int host_chunks,
                                                                 unsigned char *__restrict__ out, int xb, int cb, int tb)
{
    __shared__ unsigned lds[kItemsPerBlock * 3 + 1];
    const long long off_xyz = 12 + 8ll * n_chunks, off_rgb = off_xyz + 12ll * n_send, off_tri = off_rgb + 3ll * n_send;
    const int b = blockIdx.x;
    if (b < xb) {                                             // verticesArray (:66-75, :97)
        const int first = b * kItemsPerBlock, n = min(kItemsPerBlock, n_send - first);
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int l = threadIdx.x + 256 * q;
            if (l < n) {
                const uint4 v = verts[first + l];
                lds[3 * l] = v.y; lds[3 * l + 1] = v.z; lds[3 * l + 2] = v.w;
            }
        }
        __syncthreads();
        block_stream_store(out + off_xyz + 12ll * first, lds, 12 * n);
    } else if (b < xb + cb) {                                 // colorsArray (:69-71, :98)
        const int first = (b - xb) * kItemsPerBlock, n = min(kItemsPerBlock, n_send - first);
        const int i0 = first + 4 * threadIdx.x;
        unsigned c[4];
#pragma unroll
        for (int q = 0; q < 4; q++) c[q] = i0 + q < n_send ? (verts[i0 + q].x & 0xffffffu) : 0u;
        lds[3 * threadIdx.x + 0] = c[0] | (c[1] << 24);
        lds[3 * threadIdx.x + 1] = (c[1] >> 8) | (c[2] << 16);
        lds[3 * threadIdx.x + 2] = (c[2] >> 16) | (c[3] << 8);
        __syncthreads();
        block_stream_store(out + off_rgb + 3ll * first, lds, 3 * n);
    } else if (b < xb + cb + tb) {                            // trianglesBuffer (:80-81, :99)
        const long long first = (long long)(b - xb - cb) * (kItemsPerBlock * 3);
        const long long total = 3ll * n_tri;
        const int n = (int)min((long long)kItemsPerBlock * 3, total - first);
        for (int l = threadIdx.x; l < n; l += 256) lds[l] = (unsigned)tri[first + l];
        __syncthreads();
        block_stream_store(out + off_tri + 4 * first, lds, 4 * n);
    } else {                                                  // WriteInt x 3 + the two chunk-size arrays (:92-96)
        int *o = reinterpret_cast<int *>(out);
        if (threadIdx.x == 0) { o[0] = n_send; o[1] = n_tri; o[2] = n_chunks; }
        for (int i = threadIdx.x; i < n_chunks; i += 256) {
            int vs, ts;
            if (host_chunks) {                                // formVerticesChunks (:177-201): sizes follow from n alone
                vs = min(kChunkLimit, n_send - i * kChunkLimit);
                ts = 0;
            } else {
                vs = v_chunks[i];
                ts = t_chunks[i];
            }
            o[3 + i] = vs;
            o[3 + n_chunks + i] = ts;
        }
    }
}

}  // namespace

// =================================================================================================================================
struct LsnTransfer {
    int device = 0;
    int max_v = 0, max_t = 0;
    lsn::DevBuf tag, lidx, new_v, new_tri, bsum, boff, v_chunks, t_chunks, state;
    lsn::DevBuf l_cnt, l_uses, l_prev, l_pack, l_totals, l_rank, l_start, l_vbase, l_state;   // the link path
    ChunkState *h_state = nullptr;      // pinned
    LinkState *h_link = nullptr;        // pinned
    int last_path = 0;                  // 0: vertices only, 1: link path, 2: window walk
    int max_chunks = 0;
    int last_chunks = 0, last_send = 0;
    std::mutex mu;
    ~LsnTransfer() {
        if (h_state) (void)hipHostFree(h_state);
        if (h_link) (void)hipHostFree(h_link);
    }
};

extern "C" {

static LsnTransfer *lsnTransferCreate_impl(int device, int max_vertices, int max_triangles)
{
    lsn::clear_error();
    if (max_vertices < 0 || max_triangles < 0) { lsn::set_error("lsnTransferCreate: negative capacity"); return nullptr; }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) {
        lsn::set_error("lsnTransferCreate: no HIP device %d (this library has no CPU path)", device);
        return nullptr;
    }
    LSN_HIP_NULL(hipSetDevice(device));
    auto *t = new LsnTransfer;
    t->device = device;
    t->max_v = max_vertices;
    t->max_t = max_triangles;
    t->max_chunks = (int)(3ll * max_triangles / kChunkLimit + max_vertices / kChunkLimit + 2);
    const size_t nv = (size_t)(max_vertices > 0 ? max_vertices : 1), nt3 = (size_t)(max_triangles > 0 ? 3ll * max_triangles : 1);
    if (t->tag.reserve(nv * 8) || t->lidx.reserve(nv * 4) || t->new_v.reserve(nt3 * 16) || t->new_tri.reserve(nt3 * 4) ||
        t->bsum.reserve(kWindowBlocks * 4) || t->boff.reserve(kWindowBlocks * 4) || t->v_chunks.reserve((size_t)t->max_chunks * 4) ||
        t->t_chunks.reserve((size_t)t->max_chunks * 4) || t->state.reserve(sizeof(ChunkState)) ||
        t->l_cnt.reserve(nv * 4) || t->l_uses.reserve(nv * 4 * kMaxUses) || t->l_prev.reserve(nt3 * 4 + 32) || t->l_pack.reserve((nt3 / 3 + 8) * 8) ||
        t->l_totals.reserve((nt3 / kScanPerBlock + 2) * 8) || t->l_rank.reserve(nt3 * 4 + 32) || t->l_start.reserve(((size_t)t->max_chunks + 1) * 4) ||
        t->l_vbase.reserve(((size_t)t->max_chunks + 1) * 4) || t->l_state.reserve(sizeof(LinkState))) {
        delete t;
        return nullptr;
    }
    if (hipHostMalloc(reinterpret_cast<void **>(&t->h_link), sizeof(LinkState), hipHostMallocDefault) != hipSuccess) {
        lsn::set_error("lsnTransferCreate: hipHostMalloc failed");
        delete t;
        return nullptr;
    }
    if (hipHostMalloc(reinterpret_cast<void **>(&t->h_state), sizeof(ChunkState), hipHostMallocDefault) != hipSuccess) {
        lsn::set_error("lsnTransferCreate: hipHostMalloc failed");
        delete t;
        return nullptr;
    }
    return t;
}

LsnTransfer *lsnTransferCreate(int device, int max_vertices, int max_triangles)
{
    return lsn::guarded<LsnTransfer *>("lsnTransferCreate", static_cast<LsnTransfer *>(nullptr), [&]() { return lsnTransferCreate_impl(device, max_vertices, max_triangles); });
}

static void lsnTransferDestroy_impl(LsnTransfer *t)
{
    if (!t) return;
    (void)hipSetDevice(t->device);
    delete t;
}

void lsnTransferDestroy(LsnTransfer *t)
{
    lsn::guarded_void("lsnTransferDestroy", [&]() { lsnTransferDestroy_impl(t); });
}

static int lsnTransferLastPath_impl(LsnTransfer *t)
{
    if (!t) return -1;
    std::lock_guard<std::mutex> guard(t->mu);
    return t->last_path;
}

int lsnTransferLastPath(LsnTransfer *t)
{
    return lsn::guarded<int>("lsnTransferLastPath", static_cast<int>(-1), [&]() { return lsnTransferLastPath_impl(t); });
}

static long long lsnTransferFrameBound_impl(int n_vertices, int n_triangles)
{
    const long long nt = n_triangles > 0 ? n_triangles : 0, nv = n_vertices > 0 ? n_vertices : 0;
    const long long send = nt > 0 ? 3 * nt : nv;
    const long long chunks = 3 * nt / kChunkLimit + nv / kChunkLimit + 2;
    return 12 + 8 * chunks + 15 * send + 12 * nt;
}

long long lsnTransferFrameBound(int n_vertices, int n_triangles)
{
    return lsn::guarded<long long>("lsnTransferFrameBound", static_cast<long long>(-1), [&]() { return lsnTransferFrameBound_impl(n_vertices, n_triangles); });
}

static long long lsnTransferPack_impl(LsnTransfer *t, const void *d_vertices, int n_vertices, const int *d_triangles, int n_triangles,
                          void *d_out, long long out_cap, void *stream)
{
    lsn::clear_error();
    if (!t) { lsn::set_error("lsnTransferPack: null handle"); return -1; }
    if (n_vertices < 0 || n_triangles < 0 || n_vertices > t->max_v || n_triangles > t->max_t) {
        lsn::set_error("lsnTransferPack: %d vertices / %d triangles exceed the handle's capacity (%d / %d)", n_vertices, n_triangles, t->max_v, t->max_t);
        return -1;
    }
    if ((n_vertices && !d_vertices) || (n_triangles && !d_triangles) || !d_out) { lsn::set_error("lsnTransferPack: null buffer"); return -1; }
    if (n_triangles > 0 && n_vertices == 0) { lsn::set_error("lsnTransferPack: triangles without vertices"); return -1; }
    std::lock_guard<std::mutex> guard(t->mu);
    LSN_HIP(hipSetDevice(t->device));
    hipStream_t s = lsn::as_stream(stream);
    const uint4 *src_v = static_cast<const uint4 *>(d_vertices);
    const int *src_t = d_triangles;
    int n_send = n_vertices, n_chunks = 0, host_chunks = 1;
    bool linked = false;
    static const bool no_links = getenv("LSN_TRANSFER_WINDOW_WALK") != nullptr;       // ablation: the chunk-by-chunk walk only
    if (n_triangles > 0 && !no_links && 3ll * n_triangles / kChunkLimit + 2 <= kLinkChunksLds &&
        (n_triangles + kScanPerBlock - 1) / kScanPerBlock <= kLinkTotalsLds && t->max_chunks <= kLinkChunksLds) {
        const int n_pos = 3 * n_triangles;
        LinkState *st = t->l_state.as<LinkState>();
        int *cnt = t->l_cnt.as<int>(), *rank = t->l_rank.as<int>();
        unsigned *prev = t->l_prev.as<unsigned>();
        auto *pack = t->l_pack.as<unsigned long long>();
        auto *totals64 = t->l_totals.as<unsigned long long>();
        int *totals32 = t->l_totals.as<int>();
        LSN_HIP(hipMemsetAsync(st, 0, sizeof(LinkState), s));
        LSN_HIP(hipMemsetAsync(cnt, 0, (size_t)n_vertices * 4, s));
        link_uses_kernel<<<(n_pos + kUsesThreads * kUsesPer - 1) / (kUsesThreads * kUsesPer), kUsesThreads, 0, s>>>(src_t, n_pos, n_vertices, cnt,
                                                                                                                 t->l_uses.as<int>(), st);
        link_prev_kernel<<<(n_vertices + 255) / 256, 256, 0, s>>>(cnt, t->l_uses.as<int>(), n_vertices, prev, st);
        link_pack_kernel<<<(n_triangles + 255) / 256, 256, 0, s>>>(prev, n_triangles, pack, st);
        const int pb = (n_triangles + kScanPerBlock - 1) / kScanPerBlock;
        scan_totals_kernel<unsigned long long><<<pb, 256, 0, s>>>(pack, n_triangles, totals64);
        scan_top_kernel<unsigned long long><<<1, 1024, 0, s>>>(totals64, pb);
        scan_apply_kernel<unsigned long long><<<pb, 256, 0, s>>>(pack, n_triangles, totals64);
        link_chunks_kernel<<<1, 64, 0, s>>>(pack, totals64, n_triangles, t->max_chunks, t->l_start.as<int>(), t->l_vbase.as<int>(),
                                            t->v_chunks.as<int>(), t->t_chunks.as<int>(), st);
        const int rb = (n_pos + kScanPerBlock - 1) / kScanPerBlock;
        rank_totals_kernel<<<rb, 256, 0, s>>>(prev, n_pos, t->l_start.as<int>(), st, totals32);
        scan_top_kernel<int><<<1, 1024, 0, s>>>(totals32, rb);
        rank_apply_kernel<<<rb, 256, 0, s>>>(prev, n_pos, t->l_start.as<int>(), st, totals32, rank);
        link_emit_kernel<<<(n_pos + 255) / 256, 256, 0, s>>>(src_t, prev, n_pos, src_v, t->l_start.as<int>(), t->l_vbase.as<int>(), st, rank,
                                                             t->new_v.as<uint4>(), t->new_tri.as<int>());
        LSN_HIP(hipGetLastError());
        LSN_HIP(hipMemcpyAsync(t->h_link, st, sizeof(LinkState), hipMemcpyDeviceToHost, s));
        LSN_HIP(hipStreamSynchronize(s));
        if (t->h_link->bad) {
            lsn::set_error("lsnTransferPack: a triangle index lies outside [0, %d)", n_vertices);
            return -1;
        }
        if (!t->h_link->fallback) {
            linked = true;
            host_chunks = 0;
            n_send = t->h_link->n_send;
            n_chunks = t->h_link->n_chunks;
            src_v = t->new_v.as<uint4>();
            src_t = t->new_tri.as<int>();
            t->last_path = 1;
        }
    }
    if (linked) {
    } else if (n_triangles > 0) {
        t->last_path = 2;
        host_chunks = 0;
        ChunkState init{};
        init.p_first = init.p_last = -1;
        *t->h_state = init;
        LSN_HIP(hipMemcpyAsync(t->state.p, t->h_state, sizeof(ChunkState), hipMemcpyHostToDevice, s));
        LSN_HIP(hipMemsetAsync(t->tag.p, 0xff, (size_t)n_vertices * 8, s));
        ChunkState *st = t->state.as<ChunkState>();
        auto *tag = t->tag.as<unsigned long long>();
        const long long max_iter = n_triangles / kWindowTri + 3ll * n_triangles / kChunkLimit + 4;
        long long it = 0;
        bool done = false;
        while (!done) {
            for (int r = 0; r < 4; r++, it++) {
                chunk_tag_kernel<<<kWindowBlocks, kTriPerBlock, 0, s>>>(st, src_t, n_triangles, n_vertices, tag, t->lidx.as<int>(), t->new_tri.as<int>());
                chunk_count_kernel<<<kWindowBlocks, kTriPerBlock, 0, s>>>(st, src_t, n_triangles, n_vertices, tag, t->bsum.as<int>(), t->boff.as<int>());
                chunk_emit_kernel<<<kWindowBlocks, kTriPerBlock, 0, s>>>(st, src_t, n_triangles, n_vertices, src_v, tag, t->boff.as<int>(),
                                                                         t->lidx.as<int>(), t->new_v.as<uint4>(), t->v_chunks.as<int>(), t->t_chunks.as<int>());
            }
            LSN_HIP(hipMemcpyAsync(t->h_state, t->state.p, sizeof(ChunkState), hipMemcpyDeviceToHost, s));
            LSN_HIP(hipStreamSynchronize(s));
            done = t->h_state->done != 0;
            if (!done && it > max_iter) { lsn::set_error("lsnTransferPack: chunk search did not terminate"); return -1; }
        }
        // the new indices of the last window (idempotent when a pass after the last emit already wrote them)
        chunk_tag_kernel<<<kWindowBlocks, kTriPerBlock, 0, s>>>(st, src_t, n_triangles, n_vertices, tag, t->lidx.as<int>(), t->new_tri.as<int>());
        if (t->h_state->bad) {
            (void)hipStreamSynchronize(s);
            lsn::set_error("lsnTransferPack: a triangle index lies outside [0, %d)", n_vertices);
            return -1;
        }
        n_send = t->h_state->vbase;
        n_chunks = t->h_state->c;
        src_v = t->new_v.as<uint4>();
        src_t = t->new_tri.as<int>();
    } else {
        t->last_path = 0;
        n_chunks = (n_vertices + kChunkLimit - 1) / kChunkLimit;          // formVerticesChunks (:177-201)
    }
    const long long need = 12 + 8ll * n_chunks + 15ll * n_send + 12ll * n_triangles;
    if (need > out_cap) { lsn::set_error("lsnTransferPack: the stream needs %lld bytes, the buffer holds %lld", need, out_cap); return -1; }
    if (((size_t)d_out & 3) != 0) { lsn::set_error("lsnTransferPack: d_out must be 4-byte aligned"); return -1; }
    const int xb = (n_send + kItemsPerBlock - 1) / kItemsPerBlock, cb = xb;
    const int tb = (int)((3ll * n_triangles + kItemsPerBlock * 3 - 1) / (kItemsPerBlock * 3));
    transfer_assemble_kernel<<<xb + cb + tb + 1, 256, 0, s>>>(src_v, n_send, src_t, n_triangles, n_chunks, t->v_chunks.as<int>(), t->t_chunks.as<int>(),
                                                               host_chunks, static_cast<unsigned char *>(d_out), xb, cb, tb);
    LSN_HIP(hipGetLastError());
    LSN_HIP(hipStreamSynchronize(s));
    t->last_chunks = n_chunks;
    t->last_send = n_send;
    return need;
}

long long lsnTransferPack(LsnTransfer *t, const void *d_vertices, int n_vertices, const int *d_triangles, int n_triangles,
                          void *d_out, long long out_cap, void *stream)
{
    return lsn::guarded<long long>("lsnTransferPack", static_cast<long long>(-1), [&]() { return lsnTransferPack_impl(t, d_vertices, n_vertices, d_triangles, n_triangles, d_out, out_cap, stream); });
}

static int ply_header(PlyHeader *h, int nV, int nT)
{
    h->len = snprintf(h->text, sizeof h->text,
                      "ply\nformat binary_little_endian 1.0\r\n"          // StreamWriter.WriteLine on Windows (Utils.cs:234)
                      "element vertex %d\n"
                      "property float x\nproperty float y\nproperty float z\nproperty uchar red\nproperty uchar green\nproperty uchar blue\n"
                      "element face %d\n"
                      "property list uchar int vertex_index\n"
                      "end_header\n", nV, nT);
    return h->len > 0 && h->len < (int)sizeof h->text ? 0 : -1;
}

static long long lsnPlyBinaryBytes_impl(int n_vertices, int n_triangles)
{
    PlyHeader h;
    if (n_vertices < 0 || n_triangles < 0 || ply_header(&h, n_vertices, n_triangles)) return -1;
    return h.len + 15ll * n_vertices + 13ll * n_triangles;
}

long long lsnPlyBinaryBytes(int n_vertices, int n_triangles)
{
    return lsn::guarded<long long>("lsnPlyBinaryBytes", static_cast<long long>(-1), [&]() { return lsnPlyBinaryBytes_impl(n_vertices, n_triangles); });
}

static long long lsnPlyPack_impl(int device, const void *d_vertices, int n_vertices, const int *d_triangles, int n_triangles, void *d_out,
                     long long out_cap, void *stream)
{
    lsn::clear_error();
    PlyHeader h;
    if (n_vertices < 0 || n_triangles < 0 || ply_header(&h, n_vertices, n_triangles)) { lsn::set_error("lsnPlyPack: bad counts"); return -1; }
    if ((n_vertices && !d_vertices) || (n_triangles && !d_triangles) || !d_out) { lsn::set_error("lsnPlyPack: null buffer"); return -1; }
    const long long need = h.len + 15ll * n_vertices + 13ll * n_triangles;
    if (need > out_cap) { lsn::set_error("lsnPlyPack: the file needs %lld bytes, the buffer holds %lld", need, out_cap); return -1; }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) {
        lsn::set_error("lsnPlyPack: no HIP device %d (this library has no CPU path)", device);
        return -1;
    }
    LSN_HIP(hipSetDevice(device));
    const int vb = (n_vertices + kItemsPerBlock - 1) / kItemsPerBlock, fb = (n_triangles + kItemsPerBlock - 1) / kItemsPerBlock;
    ply_pack_kernel<<<vb + fb + 1, 256, 0, lsn::as_stream(stream)>>>(static_cast<const uint4 *>(d_vertices), n_vertices, d_triangles, n_triangles,
                                                                     static_cast<unsigned char *>(d_out), h, vb, fb);
    LSN_HIP(hipGetLastError());
    return need;
}

long long lsnPlyPack(int device, const void *d_vertices, int n_vertices, const int *d_triangles, int n_triangles, void *d_out,
                     long long out_cap, void *stream)
{
    return lsn::guarded<long long>("lsnPlyPack", static_cast<long long>(-1), [&]() { return lsnPlyPack_impl(device, d_vertices, n_vertices, d_triangles, n_triangles, d_out, out_cap, stream); });
}

}  // extern "C"

// ---- inbound: frame message + recording file (host-side parsing) ------------------------------------------------------------------
namespace {
struct Zstd {
    void *handle = nullptr;
    size_t (*decompress)(void *, size_t, const void *, size_t) = nullptr;
    unsigned long long (*decompressed_size)(const void *, size_t) = nullptr;
    unsigned (*is_error)(size_t) = nullptr;
    size_t (*compress)(void *, size_t, const void *, size_t, int) = nullptr;
    size_t (*bound)(size_t) = nullptr;
    bool ok = false;
};

Zstd &zstd()
{
    static Zstd z;
    static std::once_flag once;
    std::call_once(once, [] {
        for (const char *name : {"libzstd.so.1", "libzstd.so"}) {
            z.handle = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (z.handle) break;
        }
        if (!z.handle) return;
        z.decompress = reinterpret_cast<decltype(z.decompress)>(dlsym(z.handle, "ZSTD_decompress"));
        z.decompressed_size = reinterpret_cast<decltype(z.decompressed_size)>(dlsym(z.handle, "ZSTD_getDecompressedSize"));
        z.is_error = reinterpret_cast<decltype(z.is_error)>(dlsym(z.handle, "ZSTD_isError"));
        z.compress = reinterpret_cast<decltype(z.compress)>(dlsym(z.handle, "ZSTD_compress"));
        z.bound = reinterpret_cast<decltype(z.bound)>(dlsym(z.handle, "ZSTD_compressBound"));
        z.ok = z.decompress && z.decompressed_size && z.is_error && z.compress && z.bound;
    });
    return z;
}

int rd_i32(const unsigned char *p)
{
    int v;
    memcpy(&v, p, 4);
    return v;
}

// walks the body block (liveScanClient.cpp:233-268 / KinectSocket.cs:262-303); returns its length or -1
long long body_block_length(const unsigned char *b, long long avail, int *n_bodies)
{
    if (avail < 4) return -1;
    const int nb = rd_i32(b);
    if (nb < 0) return -1;
    long long pos = 4;
    for (int i = 0; i < nb; i++) {
        if (pos + 5 > avail) return -1;
        const int nj = rd_i32(b + pos + 1);
        if (nj < 0) return -1;
        pos += 5 + 28ll * nj;
        if (pos > avail) return -1;
    }
    *n_bodies = nb;
    return pos;
}
}  // namespace

extern "C" {

int lsnZstdAvailable(void) { return zstd().ok ? 1 : 0; }

static int lsnFrameParseHeader_impl(const unsigned char *header16, LsnFrameInfo *info)
{
    lsn::clear_error();
    if (!header16 || !info) { lsn::set_error("lsnFrameParseHeader: null argument"); return -1; }
    info->payload_bytes = rd_i32(header16);                     // KinectSocket.cs:229
    info->compressed = rd_i32(header16 + 4);                    // :237
    info->width = rd_i32(header16 + 8);                         // :238
    info->height = rd_i32(header16 + 12);                       // :239
    if (info->payload_bytes <= 0) return 1;                     // :231-235: "no more stored frames"
    if (info->width < 0 || info->height < 0 || (long long)info->width * info->height > (1ll << 26)) {
        lsn::set_error("lsnFrameParseHeader: implausible frame size %d x %d", info->width, info->height);
        return -1;
    }
    return 0;
}

int lsnFrameParseHeader(const unsigned char *header16, LsnFrameInfo *info)
{
    return lsn::guarded<int>("lsnFrameParseHeader", static_cast<int>(-1), [&]() { return lsnFrameParseHeader_impl(header16, info); });
}

static long long lsnFrameDecode_impl(const unsigned char *payload, int payload_bytes, int compressed, int width, int height,
                         unsigned char *depth_out, unsigned char *rgb_out, unsigned char *bodies_out, int bodies_cap,
                         int *n_bodies)
{
    lsn::clear_error();
    if (!payload || payload_bytes <= 0 || width < 0 || height < 0) { lsn::set_error("lsnFrameDecode: bad arguments"); return -1; }
    const long long P = (long long)width * height;
    const unsigned char *raw = payload;
    long long raw_len = payload_bytes;
    std::vector<unsigned char> tmp;
    if (compressed == 1) {                                      // KinectSocket.cs:247-248
        Zstd &z = zstd();
        if (!z.ok) { lsn::set_error("lsnFrameDecode: compressed frame but libzstd.so.1 could not be loaded"); return -1; }
        const unsigned long long out = z.decompressed_size(payload, (size_t)payload_bytes);     // ZSTDDecompressor.cs:28
        // a frame is w*h*5 bytes plus a few KB of body joints: anything else is a corrupted length field, not worth allocating
        if (out < (unsigned long long)(5 * P + 4) || out > (unsigned long long)(5 * P) + (1ull << 20)) {
            lsn::set_error("lsnFrameDecode: the zstd frame announces %llu bytes, a %d x %d frame has %lld + bodies", out, width, height, 5 * P);
            return -1;
        }
        tmp.resize((size_t)out);
        const size_t got = z.decompress(tmp.data(), tmp.size(), payload, (size_t)payload_bytes);
        if (z.is_error(got) || got != out) { lsn::set_error("lsnFrameDecode: zstd decompression failed"); return -1; }
        raw = tmp.data();
        raw_len = (long long)out;
    }
    if (raw_len < 5 * P + 4) { lsn::set_error("lsnFrameDecode: payload of %lld bytes is shorter than %d x %d x 5 + 4", raw_len, width, height); return -1; }
    int nb = 0;
    const long long bl = body_block_length(raw + 5 * P, raw_len - 5 * P, &nb);
    if (bl < 0) { lsn::set_error("lsnFrameDecode: inconsistent body block"); return -1; }
    if (depth_out) memcpy(depth_out, raw, (size_t)(2 * P));                              // KinectSocket.cs:256
    if (rgb_out) memcpy(rgb_out, raw + 2 * P, (size_t)(3 * P));                          // :257
    if (bodies_out) {
        if (bl > bodies_cap) { lsn::set_error("lsnFrameDecode: body block of %lld bytes exceeds the buffer (%d)", bl, bodies_cap); return -1; }
        memcpy(bodies_out, raw + 5 * P, (size_t)bl);
    }
    if (n_bodies) *n_bodies = nb;
    return bl;
}

long long lsnFrameDecode(const unsigned char *payload, int payload_bytes, int compressed, int width, int height,
                         unsigned char *depth_out, unsigned char *rgb_out, unsigned char *bodies_out, int bodies_cap,
                         int *n_bodies)
{
    return lsn::guarded<long long>("lsnFrameDecode", static_cast<long long>(-1), [&]() { return lsnFrameDecode_impl(payload, payload_bytes, compressed, width, height, depth_out, rgb_out, bodies_out, bodies_cap, n_bodies); });
}

static long long lsnFrameEncode_impl(const unsigned char *depth, const unsigned char *rgb, int width, int height, const unsigned char *bodies,
                         int bodies_bytes, int compression_level, unsigned char *out, long long out_cap)
{
    lsn::clear_error();
    if (!depth || !rgb || !out || width < 0 || height < 0) { lsn::set_error("lsnFrameEncode: bad arguments"); return -1; }
    static const unsigned char no_bodies[4] = {0, 0, 0, 0};
    if (!bodies || bodies_bytes < 4) { bodies = no_bodies; bodies_bytes = 4; }
    const long long P = (long long)width * height, size = 5 * P + bodies_bytes;
    if (size > 0x7fffffffll) { lsn::set_error("lsnFrameEncode: frame too large"); return -1; }
    int isize = (int)size, comp = compression_level > 0 ? 1 : 0;
    if (!comp) {
        if (16 + size > out_cap) { lsn::set_error("lsnFrameEncode: needs %lld bytes", 16 + size); return -1; }
        memcpy(out + 16, depth, (size_t)(2 * P));
        memcpy(out + 16 + 2 * P, rgb, (size_t)(3 * P));
        memcpy(out + 16 + 5 * P, bodies, (size_t)bodies_bytes);
    } else {                                                    // liveScanClient.cpp:270-281
        Zstd &z = zstd();
        if (!z.ok) { lsn::set_error("lsnFrameEncode: compression requested but libzstd.so.1 could not be loaded"); return -1; }
        std::vector<unsigned char> raw((size_t)size);
        memcpy(raw.data(), depth, (size_t)(2 * P));
        memcpy(raw.data() + 2 * P, rgb, (size_t)(3 * P));
        memcpy(raw.data() + 5 * P, bodies, (size_t)bodies_bytes);
        std::vector<unsigned char> packed(z.bound((size_t)size));
        const size_t c = z.compress(packed.data(), packed.size(), raw.data(), raw.size(), compression_level);
        if (z.is_error(c)) { lsn::set_error("lsnFrameEncode: zstd compression failed"); return -1; }
        if (16 + (long long)c > out_cap) { lsn::set_error("lsnFrameEncode: needs %lld bytes", 16 + (long long)c); return -1; }
        memcpy(out + 16, packed.data(), c);
        isize = (int)c;
    }
    memcpy(out, &isize, 4);                                     // liveScanClient.cpp:284-288
    memcpy(out + 4, &comp, 4);
    memcpy(out + 8, &width, 4);
    memcpy(out + 12, &height, 4);
    return 16 + (long long)isize;
}

long long lsnFrameEncode(const unsigned char *depth, const unsigned char *rgb, int width, int height, const unsigned char *bodies,
                         int bodies_bytes, int compression_level, unsigned char *out, long long out_cap)
{
    return lsn::guarded<long long>("lsnFrameEncode", static_cast<long long>(-1), [&]() { return lsnFrameEncode_impl(depth, rgb, width, height, bodies, bodies_bytes, compression_level, out, out_cap); });
}

static long long lsnRecordingAppend_impl(unsigned char *out, long long cap, const unsigned char *frame, int len, int timestamp_ms)
{
    lsn::clear_error();
    char hdr[96];
    const int hl = snprintf(hdr, sizeof hdr, "bufferSize= %d\nframe_timestamp= %d\n", len, timestamp_ms);   // frameFileWriterReader.cpp:123
    const long long need = hl + (long long)len + 1;
    if (!out || len < 0 || need > cap) { lsn::set_error("lsnRecordingAppend: needs %lld bytes", need); return -1; }
    memcpy(out, hdr, (size_t)hl);
    if (len > 0) memcpy(out + hl, frame, (size_t)len);
    out[hl + len] = '\n';                                       // :127
    return need;
}

long long lsnRecordingAppend(unsigned char *out, long long cap, const unsigned char *frame, int len, int timestamp_ms)
{
    return lsn::guarded<long long>("lsnRecordingAppend", static_cast<long long>(-1), [&]() { return lsnRecordingAppend_impl(out, cap, frame, len, timestamp_ms); });
}

}  // extern "C"

namespace {
bool is_space(unsigned char c) { return c == ' ' || c == '\n' || c == '\r' || c == '\t' || c == '\v' || c == '\f'; }

// one "%s" of the reader's fscanf (frameFileWriterReader.cpp:66)
bool next_token(const unsigned char *f, long long len, long long *pos, long long *b, long long *e)
{
    long long p = *pos;
    while (p < len && is_space(f[p])) p++;
    if (p >= len) return false;
    *b = p;
    while (p < len && !is_space(f[p])) p++;
    *e = p;
    *pos = p;
    return true;
}

bool next_int(const unsigned char *f, long long len, long long *pos, int *out)
{
    long long b, e;
    if (!next_token(f, len, pos, &b, &e) || e - b > 11) return false;
    char tmp[16];
    memcpy(tmp, f + b, (size_t)(e - b));
    tmp[e - b] = 0;
    char *endp;
    const long v = strtol(tmp, &endp, 10);
    if (endp == tmp || *endp != 0) return false;
    *out = (int)v;
    return true;
}
}  // namespace

extern "C" {

static long long lsnRecordingNext_impl(const unsigned char *file, long long len, long long pos, long long *frame_off, int *frame_len,
                           int *timestamp_ms)
{
    lsn::clear_error();
    if (!file || pos < 0 || !frame_off || !frame_len || !timestamp_ms) { lsn::set_error("lsnRecordingNext: bad arguments"); return -1; }
    long long b, e;
    int size = 0, ts = 0;
    if (!next_token(file, len, &pos, &b, &e)) return -1;        // end of file: not an error
    if (!next_int(file, len, &pos, &size) || !next_token(file, len, &pos, &b, &e) || !next_int(file, len, &pos, &ts) || size < 0) {
        lsn::set_error("lsnRecordingNext: malformed record header at byte %lld", b);
        return -1;
    }
    *frame_len = size;
    *timestamp_ms = ts;
    if (size == 0) { *frame_off = pos; return pos; }            // :72-73
    pos += 1;                                                   // fgetc '\n' (:75)
    if (pos + size > len) { lsn::set_error("lsnRecordingNext: record of %d bytes runs past the end of the file", size); return -1; }
    *frame_off = pos;
    pos += size;
    if (pos < len) pos += 1;                                    // fgetc '\n' (:78)
    return pos;
}

long long lsnRecordingNext(const unsigned char *file, long long len, long long pos, long long *frame_off, int *frame_len,
                           int *timestamp_ms)
{
    return lsn::guarded<long long>("lsnRecordingNext", static_cast<long long>(-1), [&]() { return lsnRecordingNext_impl(file, len, pos, frame_off, frame_len, timestamp_ms); });
}

}  // extern "C"
