// host.cpp -- host-side (CPU, scalar f64) half of the C ABI: the per-observation scalar prologues the reference
// computes in Python before touching any image.  Kept on the host on purpose: they are O(1) per observation and
// must reproduce libm-rounded f64 results bit for bit (an angle that rounds differently moves a 1/32-pixel
// warpAffine coordinate).  Everything per-pixel lives in the .hip files.
#include <cmath>
#include <cstdint>
#include <cstring>
#include <mutex>
#include <string>

#include "../../include/vlfm_amd.h"
#include "status.h"

namespace vlfm {
static thread_local std::string g_last_error;
void set_last_error(const char* msg) { g_last_error = msg ? msg : ""; }
}  // namespace vlfm

extern "C" const char* vlfm_last_error(void) { return vlfm::g_last_error.c_str(); }
extern "C" int vlfm_abi_version(void) { return 8; }

namespace {

const double kPi = 3.141592653589793238462643383279502884;  // == numpy.pi == CV_PI as a double

inline long round_half_even(double v) { return std::lrint(v); }  // cvRound; default FE_TONEAREST

// sin(deg) as OpenCV's drawing.cpp SinTable stores it: seven-decimal float literals.
float sin_deg_table(int deg) {
    static float tab[451];
    static std::once_flag once;
    std::call_once(once, [] {
        for (int i = 0; i <= 450; i++) tab[i] = (float)(std::round(std::sin(i * kPi / 180.0) * 1e7) / 1e7);
    });
    return tab[deg];
}

}  // namespace

// ------------------------------------------------------------------------------------------------ pose prologue
extern "C" int vlfm_value_map_pose_params(const double* h_tf, const double* h_yaw, const int32_t* h_env, int n,
                                          int map_size, int pixels_per_meter, int template_size,
                                          vlfm_vm_pose* h_out, int* bad_index) {
    if (!h_tf || !h_out || n < 0 || map_size <= 0 || template_size <= 0)
        return vlfm::fail(VLFM_ERR_INVALID, "value_map_pose_params: bad argument");
    const double ppm = (double)pixels_per_meter;
    const int half = template_size / 2;
    for (int k = 0; k < n; k++) {
        const double* tf = h_tf + 16 * k;
        // extract_yaw (geometry_utils.py:145-159).  A Python host passes numpy.arctan2's own result (h_yaw): NumPy's
        // SIMD arctan2 and libm's atan2 can differ by one ulp, and the reference calls NumPy here.
        const double yaw = h_yaw ? h_yaw[k] : std::atan2(tf[4], tf[0]);
        // rotate_image(curr_data, -yaw) (value_map.py:304, img_utils.py:23-25): centre (T//2, T//2) as Point2f,
        // np.degrees, then cv::getRotationMatrix2D's "angle *= CV_PI/180"
        const double degrees = (-yaw) * (180.0 / kPi);
        const double ang = degrees * (kPi / 180);
        const double alpha = std::cos(ang), beta = std::sin(ang);
        const double cx = (double)(float)half, cy = (double)(float)half;
        double M[6] = {alpha, beta, (1 - alpha) * cx - beta * cy, -beta, alpha, beta * cx + (1 - alpha) * cy};
        // cv::warpAffine without WARP_INVERSE_MAP inverts the matrix in double
        double D = M[0] * M[4] - M[1] * M[3];
        D = D != 0 ? 1. / D : 0;
        const double A11 = M[4] * D, A22 = M[0] * D;
        M[0] = A11; M[1] *= -D;
        M[3] *= -D; M[4] = A22;
        const double b1 = -M[0] * M[2] - M[1] * M[5];
        const double b2 = -M[3] * M[2] - M[4] * M[5];
        M[2] = b1; M[5] = b2;
        vlfm_vm_pose& p = h_out[k];
        std::memcpy(p.inv_affine, M, sizeof(M));
        // camera cell by truncation toward zero (value_map.py:309-313)
        const double cam_x = tf[3] / tf[15], cam_y = tf[7] / tf[15];
        const long px = (long)(cam_x * ppm) + map_size / 2;
        const long py = (long)(-cam_y * ppm) + map_size / 2;
        if (px < 0 || px >= map_size || py < 0 || py >= map_size) {
            if (bad_index) *bad_index = k;
            return vlfm::fail(VLFM_ERR_OUTSIDE_MAP, "Pixel location is outside the image.");
        }
        p.row0 = (int32_t)(px - half);
        p.col0 = (int32_t)(py - half);
        p.env = h_env ? h_env[k] : k;
        p.reserved = 0;
    }
    return VLFM_OK;
}

// ------------------------------------------------------------------------------------------------ cv2.ellipse sector
// Vertices (16.16 fixed point) of the polygon cv2.ellipse(img, (cx,cy), (r,r), 0, start, end, color, -1) fills:
// integer-rounded angles, ellipse2Poly arc in double from the float sine table, snapped to 16.16, duplicates dropped,
// centre appended for sectors.  Returns the vertex count or a negative status.
static int ellipse_sector_polygon(int cx, int cy, int radius, double start_angle, double end_angle, int64_t* out_xy,
                                  int capacity) {
    int a0 = (int)round_half_even(start_angle), a1 = (int)round_half_even(end_angle);
    const bool full = (a1 - a0) >= 360;
    if (a0 > a1) { int t = a0; a0 = a1; a1 = t; }
    while (a0 < 0) { a0 += 360; a1 += 360; }
    while (a1 > 360) { a1 -= 360; a0 -= 360; }
    if (a1 - a0 > 360) { a0 = 0; a1 = 360; }
    const int step = radius < 3 ? 90 : radius < 10 ? 30 : radius < 15 ? 18 : 5;
    const double ccx = (double)((int64_t)cx << 16), ccy = (double)((int64_t)cy << 16);
    const double axis = (double)((int64_t)radius << 16);
    int m = 0;
    int64_t last_x = -1, last_y = -1;
    for (int a = a0; a < a1 + step; a += step) {
        int t = a > a1 ? a1 : a;
        if (t < 0) t += 360;
        const double ex = axis * sin_deg_table(450 - t), ey = axis * sin_deg_table(t);
        const double vx = ccx + ex * sin_deg_table(90) - ey * sin_deg_table(0);   // rotation angle 0: alpha = 1, beta = 0
        const double vy = ccy + ex * sin_deg_table(0) + ey * sin_deg_table(90);
        int64_t qx = (int64_t)round_half_even(vx / 65536.0) << 16, qy = (int64_t)round_half_even(vy / 65536.0) << 16;
        qx += round_half_even(vx - (double)qx);
        qy += round_half_even(vy - (double)qy);
        if (qx == last_x && qy == last_y) continue;
        if (m >= capacity) return VLFM_ERR_CAPACITY;
        out_xy[2 * m] = qx; out_xy[2 * m + 1] = qy; m++;
        last_x = qx; last_y = qy;
    }
    if (!full) {
        if (m >= capacity) return VLFM_ERR_CAPACITY;
        out_xy[2 * m] = (int64_t)cx << 16; out_xy[2 * m + 1] = (int64_t)cy << 16; m++;
    }
    return m;
}

extern "C" int vlfm_fog_params_host(const int32_t* h_agent_px, const double* h_angle_cv2_deg, const double* h_rot_cs,
                                    double fov_deg, double max_line_len, const int32_t* h_env,
                                    const int32_t* h_explore, int n, vlfm_fog_params* h_out) {
    if (!h_agent_px || !h_angle_cv2_deg || !h_rot_cs || !h_out || n < 0)
        return vlfm::fail(VLFM_ERR_INVALID, "fog_params_host: bad argument");
    for (int k = 0; k < n; k++) {
        vlfm_fog_params& p = h_out[k];
        std::memset(&p, 0, sizeof(p));
        p.env = h_env ? h_env[k] : k;
        p.ax = h_agent_px[2 * k]; p.ay = h_agent_px[2 * k + 1];
        p.radius = (int)max_line_len;
        p.rot_c = h_rot_cs[2 * k]; p.rot_s = h_rot_cs[2 * k + 1];
        p.line_len = max_line_len * 1.05;
        if (h_explore && !h_explore[k]) { p.n_poly = 0; continue; }
        const int m = ellipse_sector_polygon(p.ax, p.ay, p.radius, h_angle_cv2_deg[k] - fov_deg / 2,
                                             h_angle_cv2_deg[k] + fov_deg / 2, (int64_t*)p.poly, VLFM_FOG_MAX_POLY);
        if (m < 0) return vlfm::fail(m, "fog_params_host: sector polygon does not fit VLFM_FOG_MAX_POLY");
        p.n_poly = m;
    }
    return VLFM_OK;
}

// ------------------------------------------------------------------------------------------------ confidence template
extern "C" int vlfm_cone_template_host(double fov, double max_depth, int pixels_per_meter, double min_confidence,
                                       float* h_conf, int conf_capacity, int64_t* h_poly_xy, int poly_capacity,
                                       int* n_poly) {
    if (!h_conf || !h_poly_xy || !n_poly || pixels_per_meter <= 0)
        return vlfm::fail(VLFM_ERR_INVALID, "cone_template_host: bad argument");
    const int size = (int)(max_depth * pixels_per_meter);  // value_map.py:323
    const int T = 2 * size + 1;
    if (size <= 0) return vlfm::fail(VLFM_ERR_INVALID, "cone_template_host: empty template");
    if ((long)T * T > conf_capacity) return vlfm::fail(VLFM_ERR_CAPACITY, "cone_template_host: conf capacity");
    // value_map.py:343-351 -- f64 scalar math per cell, stored as f32
    const int mid = T / 2;
    for (int r = 0; r < T; r++) {
        const double fwd = std::abs(r - mid);
        for (int c = 0; c < T; c++) {
            const double lat = std::abs(c - mid);
            double ang = std::atan2(lat, fwd);
            ang = (ang - 0) * (kPi / 2 - 0) / (fov / 2 - 0) + 0;           // remap(angle, 0, fov/2, 0, pi/2)
            double cf = std::pow(std::cos(ang), 2.0);                        // np.cos(angle) ** 2
            cf = (cf - 0) * (1 - min_confidence) / (1 - 0) + min_confidence;  // remap(conf, 0, 1, min_conf, 1)
            h_conf[(size_t)r * T + c] = (float)cf;
        }
    }
    // cv2.ellipse(mask, (size,size), (size,size), 0, -deg/2+90, deg/2+90, 1, -1)  (value_map.py:325-334)
    const double deg = fov * (180.0 / kPi);
    const int m = ellipse_sector_polygon(size, size, size, -deg / 2 + 90, deg / 2 + 90, h_poly_xy, poly_capacity);
    if (m < 0) return vlfm::fail(m, "cone_template_host: polygon capacity");
    *n_poly = m;
    return T;
}

extern "C" int vlfm_tan_table_host(double fov, int width, double* h_out) {
    if (!h_out || width <= 0) return vlfm::fail(VLFM_ERR_INVALID, "tan_table_host: bad argument");
    // np.linspace(-fov/2, fov/2, W): arange * step + start, last element pinned to stop
    const double start = -fov / 2, stop = fov / 2;
    const int div = width - 1;
    const double step = div > 0 ? (stop - start) / div : 0.0;
    for (int i = 0; i < width; i++) {
        double a = (double)i * step + start;
        if (i == width - 1 && width > 1) a = stop;
        h_out[i] = std::tan(a);
    }
    return VLFM_OK;
}

// ------------------------------------------------------------------------------------------------ disc raster
extern "C" int vlfm_disc_rows_host(int radius, int32_t* h_halfwidth) {
    if (!h_halfwidth || radius < 0) return vlfm::fail(VLFM_ERR_INVALID, "disc_rows_host: bad argument");
    // cv2.circle(mask, (r,r), r, 255, -1): midpoint circle, every iteration paints rows +-dy with half-width dx
    // and rows +-dx with half-width dy; a row keeps the widest span it ever received.
    for (int i = 0; i <= 2 * radius; i++) h_halfwidth[i] = -1;
    int err = 0, dx = radius, dy = 0, plus = 1, minus = (radius << 1) - 1;
    while (dx >= dy) {
        auto widen = [&](int row_off, int hw) {
            int32_t& slot = h_halfwidth[radius + row_off];
            if (hw > slot) slot = hw;
        };
        widen(-dy, dx); widen(dy, dx); widen(-dx, dy); widen(dx, dy);
        dy++;
        err += plus;
        plus += 2;
        const int mask = (err <= 0) - 1;
        err -= minus & mask;
        dx += mask;
        minus -= 2 & mask;
    }
    return VLFM_OK;
}

// ------------------------------------------------------------------------------------------------ kernel timing
#include <hip/hip_runtime.h>

#include <map>
#include <vector>

#include "profile.h"

namespace vlfm {
namespace {
struct KernelLog { std::vector<std::pair<hipEvent_t, hipEvent_t>> spans; long launches = 0; };
std::mutex g_prof_mu;
int g_prof_every = 0;   // 0 = off, n = bracket every n-th launch of each kernel
std::map<std::string, KernelLog> g_prof;
const size_t kMaxSpans = 8192;
}  // namespace

static thread_local ProfileScope* g_scope = nullptr;
ProfileScope* current_profile_scope() { return g_scope; }

ProfileScope::ProfileScope(const char* name, hipStream_t stream) : name_(name), stream_(stream) {
    prev_ = g_scope;
    g_scope = this;
    if (!g_prof_every) return;
    if (g_prof_every > 1) {
        std::lock_guard<std::mutex> lk(g_prof_mu);
        if (g_prof[name_].launches++ % g_prof_every != 0) return;
    }
    if (hipEventCreate(&start_) != hipSuccess || hipEventCreate(&stop_) != hipSuccess) return;
    active_ = true;
}
ProfileScope::~ProfileScope() {
    g_scope = prev_;
    if (!active_) return;
    if (!used_) { (void)hipEventDestroy(start_); (void)hipEventDestroy(stop_); return; }
    std::lock_guard<std::mutex> lk(g_prof_mu);
    KernelLog& log = g_prof[name_];
    if (log.spans.size() < kMaxSpans) log.spans.emplace_back(start_, stop_);
    else { (void)hipEventDestroy(start_); (void)hipEventDestroy(stop_); }
}
}  // namespace vlfm

extern "C" int vlfm_host_wait_mode(int blocking) {
    hipError_t e = hipSetDeviceFlags(blocking ? hipDeviceScheduleBlockingSync : hipDeviceScheduleAuto);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        vlfm::set_last_error((std::string("hipSetDeviceFlags: ") + hipGetErrorString(e)).c_str());
        return VLFM_ERR_HIP;
    }
    return VLFM_OK;
}

extern "C" int vlfm_profile_enable(int on) {
    std::lock_guard<std::mutex> lk(vlfm::g_prof_mu);
    for (auto& kv : vlfm::g_prof)
        for (auto& sp : kv.second.spans) { (void)hipEventDestroy(sp.first); (void)hipEventDestroy(sp.second); }
    vlfm::g_prof.clear();
    vlfm::g_prof_every = on > 0 ? on : 0;
    return VLFM_OK;
}

extern "C" int vlfm_profile_read(const char* kernel_name, double* mean_ms, int* launches) {
    if (!kernel_name || !mean_ms || !launches) return vlfm::fail(VLFM_ERR_INVALID, "profile_read: bad argument");
    std::lock_guard<std::mutex> lk(vlfm::g_prof_mu);
    auto it = vlfm::g_prof.find(kernel_name);
    *mean_ms = 0.0; *launches = 0;
    if (it == vlfm::g_prof.end()) return VLFM_OK;
    double total = 0.0; int n = 0;
    for (auto& sp : it->second.spans) {
        if (hipEventSynchronize(sp.second) != hipSuccess) continue;
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, sp.first, sp.second) == hipSuccess) { total += ms; n++; }
    }
    *launches = n;
    *mean_ms = n ? total / n : 0.0;
    return VLFM_OK;
}
