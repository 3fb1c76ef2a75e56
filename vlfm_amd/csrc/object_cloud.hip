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

// One workgroup: connected components of the core graph by min-label propagation (label = smallest core index reached),
// border attachment, cluster sizes, largest cluster, ordered output.  n <= 8192.
__global__ __launch_bounds__(1024) void dbscan_cluster_kernel(const unsigned long long* __restrict__ adj, const int* __restrict__ degree,
                                                              int n, int cb_count, int min_points, int* __restrict__ label,
                                                              int* __restrict__ sizes /* [n] scratch */, int* __restrict__ keep,
                                                              int* __restrict__ num_keep) {
    __shared__ int changed;
    __shared__ int best_label, best_size, out_count;
    const int tid = threadIdx.x, nth = blockDim.x;
    // label[i] = i for core points, INT_MAX otherwise
    for (int i = tid; i < n; i += nth) { label[i] = degree[i] >= min_points ? i : 0x7FFFFFFF; sizes[i] = 0; }
    __syncthreads();
    for (;;) {
        if (tid == 0) changed = 0;
        __syncthreads();
        for (int i = tid; i < n; i += nth) {
            if (degree[i] < min_points) continue;
            int best = label[i];
            const unsigned long long* row = adj + (size_t)i * cb_count;
            for (int cb = 0; cb < cb_count; cb++) {
                unsigned long long w = row[cb];
                while (w) {
                    const int j = cb * 64 + __builtin_ctzll(w);
                    w &= w - 1;
                    if (degree[j] >= min_points) { const int lj = label[j]; if (lj < best) best = lj; }
                }
            }
            if (best < label[i]) { label[i] = best; changed = 1; }  // monotone: racy reads only delay convergence
        }
        __syncthreads();
        const int again = changed;
        __syncthreads();
        if (!again) break;
    }
    // border points: the lowest-labelled neighbouring core cluster (clusters are seeded in order of their smallest core
    // index, so "lowest root index" == "first cluster to reach the point"); noise keeps INT_MAX
    for (int i = tid; i < n; i += nth) {
        if (degree[i] >= min_points) continue;
        int best = 0x7FFFFFFF;
        const unsigned long long* row = adj + (size_t)i * cb_count;
        for (int cb = 0; cb < cb_count; cb++) {
            unsigned long long w = row[cb];
            while (w) {
                const int j = cb * 64 + __builtin_ctzll(w);
                w &= w - 1;
                if (degree[j] >= min_points && label[j] < best) best = label[j];
            }
        }
        sizes[i] = best;  // staged so that border points do not feed each other
    }
    __syncthreads();
    for (int i = tid; i < n; i += nth) if (degree[i] < min_points) label[i] = sizes[i];
    __syncthreads();
    for (int i = tid; i < n; i += nth) sizes[i] = 0;
    __syncthreads();
    for (int i = tid; i < n; i += nth) if (label[i] != 0x7FFFFFFF) atomicAdd(&sizes[label[i]], 1);
    __syncthreads();
    if (tid == 0) {  // np.argmax over clusters in label order (roots ascending == labels ascending): first maximum
        int bl = -1, bs = 0;
        for (int r = 0; r < n; r++) if (sizes[r] > bs) { bs = sizes[r]; bl = r; }
        best_label = bl; best_size = bs; out_count = 0;
    }
    __syncthreads();
    if (best_label < 0) { if (tid == 0) *num_keep = 0; return; }
    if (tid == 0) {  // ordered compaction (np.where order); n is a few thousand
        int c = 0;
        for (int i = 0; i < n; i++) if (label[i] == best_label) keep[c++] = i;
        *num_keep = c;
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
