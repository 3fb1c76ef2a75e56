// object_cloud.hip -- gfx950 kernels for vlfm.mapping.ObjectPointCloudMap._extract_object_cloud
// (reference: /root/reference/vlfm/mapping/object_point_cloud_map.py:150-170, :186-212).
//
//   mask_erode_kernel      cv2.erode(mask * 255, None, iterations=k): 3x3 minimum, image border does not erode
//                          (morphologyDefaultBorderValue) -- on a bit-packed mask, one AND of nine shifted words per word
//   cloud_extract_kernel   valid_depth (0 -> 1), scale, get_point_cloud (geometry_utils.py:216-236) over the eroded mask in
//                          np.where order (row-major): per-row popcounts, an LDS scan, then an ordered expansion -> f64
//                          points (z, -x, -y)
//   dbscan_*               open3d PointCloud.cluster_dbscan(eps, min_points) [ext] for n <= ~5000 points, restated for a
//                          data-parallel machine: eps-adjacency as an n x n bit matrix (64 columns per word: a wavefront
//                          ballot), core = degree >= min_points (the point itself counts), clusters = connected components
//                          of the core-core graph labelled in order of their smallest core index (the order in which the
//                          sequential scan seeds them), border points join the lowest-labelled neighbouring cluster (the
//                          first one to reach them), everything else is noise; then the largest cluster (first maximum) in
//                          original point order (object_point_cloud_map.py:186-212).
// Latency-bound integer/f64 work on a few thousand points; no MFMA.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/vlfm_amd.h"
#include "profile.h"
#include "status.h"

namespace vlfm {

// ------------------------------------------------------------------------------------------------ mask -> bits, erosion
__global__ __launch_bounds__(256) void mask_pack_kernel(const unsigned char* __restrict__ mask, int H, int W, int hw,
                                                        unsigned* __restrict__ bits) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= H * hw) return;
    const int y = i / hw, w = i - y * hw;
    unsigned v = 0u;
    for (int b = 0; b < 32; b++) {
        const int x = w * 32 + b;
        if (x < W && mask[(size_t)y * W + x] != 0) v |= 1u << b;
    }
    bits[i] = v;
}

__global__ __launch_bounds__(256) void mask_erode_kernel(const unsigned* __restrict__ src, int H, int W, int hw,
                                                         unsigned* __restrict__ dst) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= H * hw) return;
    const int y = i / hw, w = i - y * hw;
    const unsigned tail = (W & 31) && w == hw - 1 ? ~((1u << (W & 31)) - 1u) : 0u;  // columns >= W count as set
    unsigned acc = 0xFFFFFFFFu;
    for (int dy = -1; dy <= 1; dy++) {
        const int yy = y + dy;
        if (yy < 0 || yy >= H) continue;  // rows outside the image count as set
        const unsigned* row = src + (size_t)yy * hw;
        const unsigned m = row[w] | tail;
        const unsigned left = w > 0 ? row[w - 1] >> 31 : 1u;                                   // column -1 counts as set
        const unsigned right = w + 1 < hw ? (row[w + 1] | ((W & 31) && w + 1 == hw - 1 ? ~((1u << (W & 31)) - 1u) : 0u)) & 1u : 1u;
        acc &= m & ((m << 1) | left) & ((m >> 1) | (right << 31));
    }
    dst[i] = acc & ~tail;
}

// ------------------------------------------------------------------------------------------------ masked unprojection
__global__ __launch_bounds__(1024) void cloud_extract_kernel(const float* __restrict__ depth, const unsigned* __restrict__ bits,
                                                             int H, int W, int hw, float scale, float offset, double fx,
                                                             double fy, double* __restrict__ cloud, int cap,
                                                             int* __restrict__ count) {
    extern __shared__ int row_off[];  // [H + 1]
    const int tid = threadIdx.x, nth = blockDim.x;
    for (int y = tid; y < H; y += nth) {
        int c = 0;
        for (int w = 0; w < hw; w++) c += __popc(bits[(size_t)y * hw + w]);
        row_off[y + 1] = c;
    }
    if (tid == 0) row_off[0] = 0;
    __syncthreads();
    if (tid == 0) {  // H <= a few thousand: a serial scan is microseconds
        for (int y = 0; y < H; y++) row_off[y + 1] += row_off[y];
        *count = row_off[H];
    }
    __syncthreads();
    // one wavefront per row: lanes own words, ranks come from a wave prefix over the word popcounts
    const int lane = tid & 63, wave = tid >> 6, n_waves = nth >> 6;
    for (int y = wave; y < H; y += n_waves) {
        int base = row_off[y];
        for (int w0 = 0; w0 < hw; w0 += 64) {
            const int w = w0 + lane;
            unsigned v = w < hw ? bits[(size_t)y * hw + w] : 0u;
            int pc = __popc(v), incl = pc;
            for (int off = 1; off < 64; off <<= 1) {
                const int t = __shfl_up(incl, off, 64);
                if (lane >= off) incl += t;
            }
            int rank = base + incl - pc;
            while (v) {
                const int b = __builtin_ctz(v);
                v &= v - 1;
                const int u = w * 32 + b;
                if (rank < cap) {
                    float d = depth[(size_t)y * W + u];
                    if (d == 0.0f) d = 1.0f;                                   // holes are "far" (:160-161)
                    const float z = __fadd_rn(__fmul_rn(d, scale), offset);    // f32 (:162)
                    const double zd = (double)z;
                    const double xc = __ddiv_rn(__dmul_rn((double)(u - W / 2), zd), fx);   // geometry_utils.py:230-232
                    const double yc = __ddiv_rn(__dmul_rn((double)(y - H / 2), zd), fy);
                    cloud[(size_t)rank * 3 + 0] = zd;
                    cloud[(size_t)rank * 3 + 1] = -xc;
                    cloud[(size_t)rank * 3 + 2] = -yc;
                }
                rank++;
            }
            base += __shfl(incl, 63, 64);
        }
    }
}

// ------------------------------------------------------------------------------------------------ DBSCAN
// adjacency: word (i, cb) bit j = |p_i - p_(64 cb + j)|^2 < eps^2 (strict: nanoflann's radius search); one wavefront per
// (row, column block): lane j tests point 64 cb + j, the ballot is the word.
__global__ __launch_bounds__(256) void dbscan_adjacency_kernel(const double* __restrict__ pts, int n, double eps2,
                                                               unsigned long long* __restrict__ adj, int cb_count,
                                                               int* __restrict__ degree) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = blockIdx.x * 4 + wave;
    if (i >= n) return;
    const double xi = pts[(size_t)i * 3], yi = pts[(size_t)i * 3 + 1], zi = pts[(size_t)i * 3 + 2];
    int deg = 0;
    for (int cb = 0; cb < cb_count; cb++) {
        const int j = cb * 64 + lane;
        bool near = false;
        if (j < n) {
            const double dx = pts[(size_t)j * 3] - xi, dy = pts[(size_t)j * 3 + 1] - yi, dz = pts[(size_t)j * 3 + 2] - zi;
            near = __dadd_rn(__dadd_rn(__dmul_rn(dx, dx), __dmul_rn(dy, dy)), __dmul_rn(dz, dz)) < eps2;
        }
        const unsigned long long word = __ballot(near);
        if (lane == 0) adj[(size_t)i * cb_count + cb] = word;
        deg += __popcll(word);
    }
    if (lane == 0) degree[i] = deg;
}

// One workgroup: connected components of the core graph by BIT-PARALLEL breadth-first search (round 4; before: min-label propagation
// that enumerated every neighbour of every core point once per sweep -- 5000 points of one dense object blob have ~1000 neighbours each,
// and with the serial size / compaction loops of one lane a call took 20-35 ms, which made the object-map stage THE cost of the
// batched full step).  The eps-adjacency is already a bit matrix; a search level is "OR the adjacency rows of the frontier": every
// wavefront takes frontier nodes, its lanes OR the row's words (masked by the core set) into registers, and the wavefront's result
// goes to LDS once per level -- n x n / 64 word operations per component in total, no neighbour enumeration.  Components are seeded
// in ascending order of their smallest core index (the order in which the sequential scan of the reference's Open3D seeds them), so
// label = that index, as before.  Border points have fewer than min_points neighbours: enumerating those is cheap.  Cluster sizes,
// the first largest cluster and the ordered index list are parallel reductions / prefix sums.  n <= 8192.
constexpr int DB_WORDS = 128;    // 8192 / 64

__global__ __launch_bounds__(1024) void dbscan_cluster_kernel(const unsigned long long* __restrict__ adj, const int* __restrict__ degree,
                                                              int n, int cb_count, int min_points, int* __restrict__ label,
                                                              int* __restrict__ sizes /* [n] scratch */, int* __restrict__ keep,
                                                              int* __restrict__ num_keep) {
    __shared__ unsigned long long core[DB_WORDS], unvis[DB_WORDS], comp[DB_WORDS], front[DB_WORDS], nxt[DB_WORDS];
    __shared__ int sh_root, sh_more, best_label, best_size;
    __shared__ int wprefix[DB_WORDS + 1];
    __shared__ unsigned long long red_key[16];
    const int tid = threadIdx.x, nth = blockDim.x, lane = tid & 63, wave = tid >> 6, nwaves = nth >> 6;
    // ---- core set as a bit set; every label starts as noise
    for (int base = 0; base < cb_count * 64; base += nth) {
        const int i = base + tid;
        const bool is_core = i < n && degree[i] >= min_points;
        const unsigned long long w = __ballot(is_core);
        if (lane == 0 && (i >> 6) < cb_count) { core[i >> 6] = w; unvis[i >> 6] = w; }
        if (i < n) { label[i] = 0x7FFFFFFF; sizes[i] = 0; }
    }
    __syncthreads();
    for (;;) {
        // ---- next seed: the smallest core index not yet in a component
        if (wave == 0) {
            int first = 0x7FFFFFFF;
            for (int wd = lane; wd < cb_count; wd += 64)
                if (unvis[wd] && first == 0x7FFFFFFF) first = wd * 64 + __builtin_ctzll(unvis[wd]);
            for (int off = 32; off > 0; off >>= 1) first = min(first, __shfl_xor(first, off, 64));
            if (lane == 0) sh_root = first;
        }
        __syncthreads();
        const int root = sh_root;
        if (root == 0x7FFFFFFF) break;
        for (int wd = tid; wd < cb_count; wd += nth) {
            const unsigned long long b = wd == (root >> 6) ? (1ull << (root & 63)) : 0ull;
            comp[wd] = b; front[wd] = b; nxt[wd] = 0ull;
        }
        __syncthreads();
        // ---- breadth-first levels
        for (;;) {
            // lane l of a wavefront accumulates words l and l + 64 of the rows of the frontier nodes its wavefront takes
            unsigned long long a0 = 0ull, a1 = 0ull;
            for (int wd = wave; wd < cb_count; wd += nwaves) {
                unsigned long long f = front[wd];          // wave-uniform
                while (f) {
                    const int i = wd * 64 + __builtin_ctzll(f);
                    f &= f - 1;
                    const unsigned long long* row = adj + (size_t)i * cb_count;
                    if (lane < cb_count) a0 |= row[lane];
                    if (lane + 64 < cb_count) a1 |= row[lane + 64];
                }
            }
            if (lane < cb_count && a0) atomicOr(&nxt[lane], a0 & core[lane]);
            if (lane + 64 < cb_count && a1) atomicOr(&nxt[lane + 64], a1 & core[lane + 64]);
            if (tid == 0) sh_more = 0;
            __syncthreads();
            for (int wd = tid; wd < cb_count; wd += nth) {
                const unsigned long long fresh = nxt[wd] & ~comp[wd];
                comp[wd] |= fresh; front[wd] = fresh; nxt[wd] = 0ull;
                if (fresh) sh_more = 1;
            }
            __syncthreads();
            if (!sh_more) break;
            __syncthreads();
        }
        // ---- the component's label, and out of the unvisited set
        for (int i = tid; i < cb_count * 64; i += nth)
            if (i < n && ((comp[i >> 6] >> (i & 63)) & 1ull)) label[i] = root;
        for (int wd = tid; wd < cb_count; wd += nth) unvis[wd] &= ~comp[wd];
        __syncthreads();
    }
    __threadfence_block();
    __syncthreads();
    // ---- border points: the lowest-labelled neighbouring core cluster (clusters are seeded in order of their smallest core index, so
    // "lowest root index" == "first cluster to reach the point"); noise keeps INT_MAX.  Read core labels only: border points do not feed
    // each other (their own labels are written after the barrier)
    for (int i = tid; i < n; i += nth) {
        int best = 0x7FFFFFFF;
        if (degree[i] < min_points) {
            const unsigned long long* row = adj + (size_t)i * cb_count;
            for (int cb = 0; cb < cb_count; cb++) {
                unsigned long long w = row[cb] & core[cb];
                while (w) {
                    const int j = cb * 64 + __builtin_ctzll(w);
                    w &= w - 1;
                    best = min(best, label[j]);
                }
            }
        }
        sizes[i] = best;   // staged
    }
    __syncthreads();
    for (int i = tid; i < n; i += nth) if (degree[i] < min_points) label[i] = sizes[i];
    __syncthreads();
    for (int i = tid; i < n; i += nth) sizes[i] = 0;
    __syncthreads();
    for (int i = tid; i < n; i += nth) if (label[i] != 0x7FFFFFFF) atomicAdd(&sizes[label[i]], 1);
    __syncthreads();
    // ---- np.argmax over clusters in label order (roots ascending == labels ascending): the FIRST maximum = largest size, then the
    // smallest root.  key = size << 32 | (0xFFFFFFFF - root): the maximum key wins
    unsigned long long key = 0ull;
    for (int r = tid; r < n; r += nth) {
        const int sz = sizes[r];
        if (sz > 0) { const unsigned long long k = ((unsigned long long)(unsigned)sz << 32) | (0xFFFFFFFFu - (unsigned)r); key = k > key ? k : key; }
    }
    for (int off = 32; off > 0; off >>= 1) { const unsigned long long o = __shfl_xor(key, off, 64); key = o > key ? o : key; }
    if (lane == 0) red_key[wave] = key;
    __syncthreads();
    if (tid == 0) {
        unsigned long long k = 0ull;
        for (int w = 0; w < nwaves; w++) k = red_key[w] > k ? red_key[w] : k;
        best_size = (int)(k >> 32);
        best_label = k ? (int)(0xFFFFFFFFu - (unsigned)(k & 0xFFFFFFFFull)) : -1;
    }
    __syncthreads();
    if (best_label < 0) { if (tid == 0) *num_keep = 0; return; }
    // ---- ordered compaction (np.where order): membership bit set -> per-word prefix -> every member writes its own slot
    const int bl = best_label;
    for (int base = 0; base < cb_count * 64; base += nth) {
        const int i = base + tid;
        const unsigned long long w = __ballot(i < n && label[i] == bl);
        if (lane == 0 && (i >> 6) < cb_count) comp[i >> 6] = w;
    }
    __syncthreads();
    if (tid == 0) {
        int c = 0;
        for (int wd = 0; wd < cb_count; wd++) { wprefix[wd] = c; c += __popcll(comp[wd]); }   // <= 128 LDS words
        wprefix[cb_count] = c;
        *num_keep = c;
    }
    __syncthreads();
    for (int i = tid; i < n; i += nth) {
        const unsigned long long w = comp[i >> 6];
        if ((w >> (i & 63)) & 1ull) keep[wprefix[i >> 6] + __popcll(w & ((1ull << (i & 63)) - 1ull))] = i;
    }
}

}  // namespace vlfm

using namespace vlfm;

extern "C" size_t vlfm_object_cloud_scratch_bytes(int height, int width) {
    if (height <= 0 || width <= 0) return 0;
    return (size_t)2 * height * ((width + 31) / 32) * sizeof(uint32_t);
}

extern "C" int vlfm_object_cloud_extract(const float* d_depth, const uint8_t* d_mask, int height, int width,
                                         int erosion_iterations, double min_depth, double max_depth, double fx, double fy,
                                         void* d_scratch, double* d_cloud, int capacity, int32_t* d_count, void* stream) {
    if (!d_depth || !d_mask || !d_scratch || !d_cloud || !d_count || height <= 0 || width <= 0 || erosion_iterations < 0 ||
        capacity <= 0 || height > 8192)
        return fail(VLFM_ERR_INVALID, "object_cloud_extract: bad argument");
    const int hw = (width + 31) / 32, words = height * hw;
    unsigned* a = (unsigned*)d_scratch;
    unsigned* b = a + words;
    VLFM_KLAUNCH(mask_pack_kernel, dim3((words + 255) / 256), dim3(256), 0, stream, d_mask, height, width, hw, a);
    for (int k = 0; k < erosion_iterations; k++) {
        VLFM_KLAUNCH(mask_erode_kernel, dim3((words + 255) / 256), dim3(256), 0, stream, a, height, width, hw, b);
        unsigned* t = a; a = b; b = t;
    }
    // NumPy: f32 image * Python float -> f32 (object_point_cloud_map.py:162)
    VLFM_TIMED("cloud_extract_kernel", stream);
    VLFM_KLAUNCH(cloud_extract_kernel, dim3(1), dim3(1024), (size_t)(height + 1) * sizeof(int), stream, d_depth, a, height,
                 width, hw, (float)(max_depth - min_depth), (float)min_depth, fx, fy, d_cloud, capacity, d_count);
    return check_launch("cloud_extract_kernel");
}

extern "C" size_t vlfm_dbscan_scratch_bytes(int n) {
    if (n <= 0) return 0;
    const size_t cb = ((size_t)n + 63) / 64;
    return (size_t)n * cb * 8 + (size_t)2 * n * sizeof(int32_t) + 256;
}

extern "C" int vlfm_dbscan_largest_cluster(const double* d_points, int n, double eps, int min_points, void* d_scratch,
                                           size_t scratch_bytes, int32_t* d_labels, int32_t* d_keep, int32_t* d_num_keep,
                                           void* stream) {
    if (!d_num_keep) return fail(VLFM_ERR_INVALID, "dbscan: d_num_keep is null");
    if (n == 0) return hipMemsetAsync(d_num_keep, 0, sizeof(int32_t), (hipStream_t)stream) == hipSuccess ? VLFM_OK : VLFM_ERR_HIP;
    if (!d_points || !d_scratch || !d_labels || !d_keep || n < 0 || n > 8192 || min_points < 1 || !(eps > 0))
        return fail(VLFM_ERR_INVALID, "dbscan: bad argument (n <= 8192)");
    if (scratch_bytes < vlfm_dbscan_scratch_bytes(n)) return fail(VLFM_ERR_CAPACITY, "dbscan: scratch too small");
    const int cb = (n + 63) / 64;
    unsigned long long* adj = (unsigned long long*)d_scratch;
    int* degree = (int*)((unsigned char*)d_scratch + (size_t)n * cb * 8);
    int* sizes = degree + n;
    {
        VLFM_TIMED("dbscan_adjacency_kernel", stream);
        VLFM_KLAUNCH(dbscan_adjacency_kernel, dim3((n + 3) / 4), dim3(256), 0, stream, d_points, n, eps * eps, adj, cb, degree);
    }
    VLFM_TIMED("dbscan_cluster_kernel", stream);
    VLFM_KLAUNCH(dbscan_cluster_kernel, dim3(1), dim3(1024), 0, stream, (const unsigned long long*)adj, (const int*)degree, n,
                 cb, min_points, d_labels, sizes, d_keep, d_num_keep);
    return check_launch("dbscan_cluster_kernel");
}
