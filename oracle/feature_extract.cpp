// TEST INFRASTRUCTURE ONLY (see oracle/linalg.hpp header). extractCloud and match*PointFromMap: PINNED (round 2) against the reference's OWN SOURCE LINES compiled over a shim (oracle/ref/, tests/test_oracle_ref_pin.py);
// the Eigen / FLANN arithmetic they call (linalg.hpp, kdtree.hpp) stays a restatement: PARITY UNPINNED for that part.
// See feature_extract.hpp for the list of reference functions restated here.
#include "feature_extract.hpp"
#include "linalg.hpp"
#include <algorithm>
#include <cmath>
#include <cfloat>
#include <cstring>

namespace orc {

namespace {
// feature_extract.hpp:48-53 compObject
struct CompObject {
    const float *cloud_curvature;
    bool operator()(int i, int j) const { return cloud_curvature[i] < cloud_curvature[j]; }
};

// The reference's comparator looks at the curvature only (hpp:48-53), so the order std::sort leaves EQUAL curvatures in is whatever
// libstdc++'s introsort does, and a NaN curvature (a non-finite input point) breaks the strict weak ordering std::sort requires
// (undefined behaviour). Tie rule 1 refines the comparator into a total order -- (curvature, index) ascending, NaN above every number --
// every outcome of which is ALSO a valid outcome of the reference's comparator on finite data; it is the order the HIP kernel sorts by
// (extract.hip: sort_sector) and the one INTEGRATION.md documents. Rule 0 (default) is the reference's comparator verbatim.
struct CompObjectTotal {
    const float *cloud_curvature;
    static unsigned long long key(float c, int i)
    {
        unsigned b;
        std::memcpy(&b, &c, sizeof(b));
        if (c != c) b = 0x7fc00000u;
        return ((unsigned long long)b << 32) | (unsigned)i;
    }
    bool operator()(int i, int j) const { return key(cloud_curvature[i], i) < key(cloud_curvature[j], j); }
};
int g_tie_rule = 0;

// the neighbour-suppression loops, feature_extract.cpp:192-213 / 233-254
inline void suppress_neighbours(const PointI *p, int ind, int *picked)
{
    for (int l = 1; l <= 5; l++) {
        float diff_x = p[ind + l].x - p[ind + l - 1].x;
        float diff_y = p[ind + l].y - p[ind + l - 1].y;
        float diff_z = p[ind + l].z - p[ind + l - 1].z;
        if (diff_x * diff_x + diff_y * diff_y + diff_z * diff_z > 0.05) break;
        picked[ind + l] = 1;
    }
    for (int l = -1; l >= -5; l--) {
        float diff_x = p[ind + l].x - p[ind + l + 1].x;
        float diff_y = p[ind + l].y - p[ind + l + 1].y;
        float diff_z = p[ind + l].z - p[ind + l + 1].z;
        if (diff_x * diff_x + diff_y * diff_y + diff_z * diff_z > 0.05) break;
        picked[ind + l] = 1;
    }
}
}  // namespace

void set_tie_rule(int rule) { g_tie_rule = rule; }

void extract_cloud(const PointI *p, int n, const int *scan_start, const int *scan_end, int n_scans, ExtractResult &out)
{
    out = ExtractResult();
    out.curvature.assign(n, 0.f);
    out.label.assign(n, 0);
    out.picked.assign(n, 0);
    std::vector<int> sort_ind(n);
    for (int i = 0; i < n; ++i) sort_ind[i] = i;
    float *cloud_curvature = out.curvature.data();
    int *cloud_label = out.label.data();
    int *cloud_neighbor_picked = out.picked.data();

    // cpp:133-142 (the reference loops i in [5, n-5) with size_t arithmetic; n < 11 is UB there, a no-op here)
    for (int i = 5; i < n - 5; i++) {
        float diff_x = p[i - 5].x + p[i - 4].x + p[i - 3].x + p[i - 2].x + p[i - 1].x - 10 * p[i].x + p[i + 1].x + p[i + 2].x + p[i + 3].x + p[i + 4].x + p[i + 5].x;
        float diff_y = p[i - 5].y + p[i - 4].y + p[i - 3].y + p[i - 2].y + p[i - 1].y - 10 * p[i].y + p[i + 1].y + p[i + 2].y + p[i + 3].y + p[i + 4].y + p[i + 5].y;
        float diff_z = p[i - 5].z + p[i - 4].z + p[i - 3].z + p[i - 2].z + p[i - 1].z - 10 * p[i].z + p[i + 1].z + p[i + 2].z + p[i + 3].z + p[i + 4].z + p[i + 5].z;
        cloud_curvature[i] = diff_x * diff_x + diff_y * diff_y + diff_z * diff_z;
    }

    CompObject comp_object{cloud_curvature};
    out.less_flat_raw_ring_off.push_back(0);
    for (int i = 0; i < n_scans; i++) {
        if (scan_end[i] - scan_start[i] < 6) { out.less_flat_raw_ring_off.push_back((int)out.less_flat_raw.size()); continue; }
        std::vector<PointI> surf_points_less_flat_scan;
        for (int j = 0; j < 6; j++) {
            int sp = scan_start[i] + (scan_end[i] - scan_start[i]) * j / 6;
            int ep = scan_start[i] + (scan_end[i] - scan_start[i]) * (j + 1) / 6 - 1;
            if (g_tie_rule == 1) std::sort(sort_ind.begin() + sp, sort_ind.begin() + ep + 1, CompObjectTotal{cloud_curvature});
            else std::sort(sort_ind.begin() + sp, sort_ind.begin() + ep + 1, comp_object);
            for (int k = sp; k < ep; ++k)
                if (cloud_curvature[sort_ind[k]] == cloud_curvature[sort_ind[k + 1]]) out.n_ties++;

            int largest_picked_num = 0;
            for (int k = ep; k >= sp; k--) {
                int ind = sort_ind[k];
                if (cloud_neighbor_picked[ind] == 0 && cloud_curvature[ind] > 0.1) {
                    largest_picked_num++;
                    if (largest_picked_num <= 2) {
                        cloud_label[ind] = 2;
                        out.sharp.push_back(ind);
                        out.less_sharp.push_back(ind);
                    } else if (largest_picked_num <= 20) {
                        cloud_label[ind] = 1;
                        out.less_sharp.push_back(ind);
                    } else {
                        break;
                    }
                    cloud_neighbor_picked[ind] = 1;
                    suppress_neighbours(p, ind, cloud_neighbor_picked);
                }
            }

            int smallest_picked_num = 0;
            for (int k = sp; k <= ep; k++) {
                int ind = sort_ind[k];
                if (cloud_neighbor_picked[ind] == 0 && cloud_curvature[ind] < 0.1) {
                    cloud_label[ind] = -1;
                    out.flat.push_back(ind);
                    smallest_picked_num++;
                    if (smallest_picked_num >= 4) break;
                    cloud_neighbor_picked[ind] = 1;
                    suppress_neighbours(p, ind, cloud_neighbor_picked);
                }
            }

            for (int k = sp; k <= ep; k++) {
                if (cloud_label[k] <= 0) {
                    surf_points_less_flat_scan.push_back(p[k]);
                    out.less_flat_raw.push_back(k);
                }
            }
        }
        out.less_flat_raw_ring_off.push_back((int)out.less_flat_raw.size());
        std::vector<PointI> ds;
        voxel_grid_xyzi(surf_points_less_flat_scan.data(), (int)surf_points_less_flat_scan.size(), 0.2f, ds);
        out.less_flat_ds.insert(out.less_flat_ds.end(), ds.begin(), ds.end());
    }
}

void voxel_grid_xyzi(const PointI *in, int n, float leaf, std::vector<PointI> &out)
{
    out.clear();
    if (n <= 0) return;
    const float inv = 1.0f / leaf;
    float min_p[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, max_p[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
    for (int i = 0; i < n; ++i) {
        const float v[3] = {in[i].x, in[i].y, in[i].z};
        if (!std::isfinite(v[0]) || !std::isfinite(v[1]) || !std::isfinite(v[2])) continue;
        for (int d = 0; d < 3; ++d) { min_p[d] = std::min(min_p[d], v[d]); max_p[d] = std::max(max_p[d], v[d]); }
    }
    int64_t dx = int64_t((max_p[0] - min_p[0]) * inv) + 1;
    int64_t dy = int64_t((max_p[1] - min_p[1]) * inv) + 1;
    int64_t dz = int64_t((max_p[2] - min_p[2]) * inv) + 1;
    if (dx * dy * dz > int64_t(INT32_MAX)) { out.assign(in, in + n); return; }   // "leaf size too small": output = input
    int min_b[3], max_b[3], div_b[3], divb_mul[3];
    for (int d = 0; d < 3; ++d) {
        min_b[d] = int(std::floor(min_p[d] * inv));
        max_b[d] = int(std::floor(max_p[d] * inv));
        div_b[d] = max_b[d] - min_b[d] + 1;
    }
    divb_mul[0] = 1; divb_mul[1] = div_b[0]; divb_mul[2] = div_b[0] * div_b[1];

    struct IdxPt {
        unsigned int idx; unsigned int cloud_point_index;
        bool operator<(const IdxPt &o) const { return idx < o.idx; }
    };
    std::vector<IdxPt> index_vector;
    index_vector.reserve(n);
    for (int i = 0; i < n; ++i) {
        if (!std::isfinite(in[i].x) || !std::isfinite(in[i].y) || !std::isfinite(in[i].z)) continue;
        int ijk0 = int(std::floor(in[i].x * inv) - float(min_b[0]));
        int ijk1 = int(std::floor(in[i].y * inv) - float(min_b[1]));
        int ijk2 = int(std::floor(in[i].z * inv) - float(min_b[2]));
        int idx = ijk0 * divb_mul[0] + ijk1 * divb_mul[1] + ijk2 * divb_mul[2];
        index_vector.push_back({(unsigned)idx, (unsigned)i});
    }
    std::sort(index_vector.begin(), index_vector.end(), std::less<IdxPt>());
    size_t i = 0;
    while (i < index_vector.size()) {
        size_t j = i + 1;
        while (j < index_vector.size() && index_vector[j].idx == index_vector[i].idx) ++j;
        // CentroidPoint<PointXYZI>: AccumulatorXYZ (Vector3f sum / n) + AccumulatorIntensity (float sum / n)
        float sx = 0.f, sy = 0.f, sz = 0.f, si = 0.f;
        for (size_t l = i; l < j; ++l) {
            const PointI &q = in[index_vector[l].cloud_point_index];
            sx += q.x; sy += q.y; sz += q.z; si += q.intensity;
        }
        float cnt = float(j - i);
        out.push_back({sx / cnt, sy / cnt, sz / cnt, si / cnt});
        i = j;
    }
}

void point_associate_to_map(const float pi[3], float po[3], const Pose &pose)
{
    Vec3d pc{double(pi[0]), double(pi[1]), double(pi[2])};
    Vec3d r = quat_rotate(pose.q, pc);
    po[0] = float(r.x + pose.t.x);
    po[1] = float(r.y + pose.t.y);
    po[2] = float(r.z + pose.t.z);
}

namespace {
inline bool check_fov_fn(const Pose &pose_local, const float point_sel[3])
{
    // feature_extract.hpp:696-715 (identical at 441-460, 586-605, 841-858)
    const float zaxis[3] = {0.0f, 0.0f, 10.0f};
    float zt[3];
    point_associate_to_map(zaxis, zt, pose_local);
    double a0 = pose_local.t.x - point_sel[0], a1 = pose_local.t.y - point_sel[1], a2 = pose_local.t.z - point_sel[2];
    float squared_side1 = float(a0 * a0 + a1 * a1 + a2 * a2);            // sqrSum<double>, stored to float
    float b0 = zt[0] - point_sel[0], b1 = zt[1] - point_sel[1], b2 = zt[2] - point_sel[2];
    float squared_side2 = b0 * b0 + b1 * b1 + b2 * b2;                    // sqrSum<float>
    float check1 = 100.0f + squared_side1 - squared_side2 - 10.0f * std::sqrt(3.0f) * std::sqrt(squared_side1);
    float check2 = 100.0f + squared_side1 - squared_side2 + 10.0f * std::sqrt(3.0f) * std::sqrt(squared_side1);
    return check1 < 0 && check2 > 0;
}
}  // namespace

bool match_corner_point_from_map(const MapCloud &map, const float *point_ori, const Pose &pose_local, Feature &feature,
                                 size_t idx, int n_neigh, bool check_fov, const MatchParams &mp)
{
    int point_search_idx[16];
    float point_search_sq_dis[16];
    const int num_neighbors = n_neigh;
    float point_sel[3];
    point_associate_to_map(point_ori, point_sel, pose_local);
    int found = map.tree.knn(point_sel, num_neighbors, point_search_idx, point_search_sq_dis);
    if (found < num_neighbors) return false;   // PCL would leave stale zeros; maps here always hold > k points
    if (point_search_sq_dis[num_neighbors - 1] < mp.min_match_sq_dis) {
        float near[16][3];
        float center[3] = {0.f, 0.f, 0.f};
        for (int j = 0; j < num_neighbors; j++) {
            const float *q = map.pts + size_t(point_search_idx[j]) * map.stride;
            near[j][0] = q[0]; near[j][1] = q[1]; near[j][2] = q[2];
            center[0] += q[0]; center[1] += q[1]; center[2] += q[2];
        }
        const float kf = float(1.0 * num_neighbors);
        center[0] /= kf; center[1] /= kf; center[2] /= kf;
        float cov_mat[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
        for (int j = 0; j < num_neighbors; j++) {
            float t[3] = {near[j][0] - center[0], near[j][1] - center[1], near[j][2] - center[2]};
            for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) cov_mat[r][c] += t[r] * t[c];
        }
        Eig3f es = eig3_sym_f(cov_mat);
        float unit_direction[3] = {es.vec[0][2], es.vec[1][2], es.vec[2][2]};
        if (es.val[2] > 3 * es.val[1]) {
            bool is_in_laser_fov = check_fov ? check_fov_fn(pose_local, point_sel) : true;
            if (is_in_laser_fov) {
                for (int d = 0; d < 3; ++d) {
                    float X1 = 0.1f * unit_direction[d] + center[d];
                    float X2 = -0.1f * unit_direction[d] + center[d];
                    feature.coeffs[d] = X1;
                    feature.coeffs[3 + d] = X2;
                }
                feature.idx = idx;
                feature.point[0] = point_ori[0]; feature.point[1] = point_ori[1]; feature.point[2] = point_ori[2];
                feature.laser_idx = (size_t)point_ori[3];
                feature.type = 'c';
                return true;
            }
        }
    }
    return false;
}

bool match_surf_point_from_map(const MapCloud &map, const float *point_ori, const Pose &pose_local, Feature &feature,
                               size_t idx, int n_neigh, bool check_fov, const MatchParams &mp)
{
    int point_search_idx[16];
    float point_search_sq_dis[16];
    const int num_neighbors = n_neigh;
    float point_sel[3];
    point_associate_to_map(point_ori, point_sel, pose_local);
    int found = map.tree.knn(point_sel, num_neighbors, point_search_idx, point_search_sq_dis);
    if (found < num_neighbors) return false;
    if (point_search_sq_dis[num_neighbors - 1] < mp.min_match_sq_dis) {
        float mat_A[16 * 3], mat_B[16];
        for (int j = 0; j < num_neighbors; j++) {
            const float *q = map.pts + size_t(point_search_idx[j]) * map.stride;
            mat_A[j * 3 + 0] = q[0]; mat_A[j * 3 + 1] = q[1]; mat_A[j * 3 + 2] = q[2];
            mat_B[j] = -1.f;
        }
        float norm[3];
        colpiv_qr_solve_f(mat_A, mat_B, num_neighbors, norm);
        float nn = std::sqrt(norm[0] * norm[0] + norm[1] * norm[1] + norm[2] * norm[2]);
        float negative_OA_dot_norm = 1 / nn;
        {   // norm.normalize()
            float z = norm[0] * norm[0] + norm[1] * norm[1] + norm[2] * norm[2];
            if (z > 0.f) { float s = std::sqrt(z); norm[0] /= s; norm[1] /= s; norm[2] /= s; }
        }
        bool plane_valid = true;
        for (int j = 0; j < num_neighbors; j++) {
            const float *q = map.pts + size_t(point_search_idx[j]) * map.stride;
            if (std::fabs(norm[0] * q[0] + norm[1] * q[1] + norm[2] * q[2] + negative_OA_dot_norm) > mp.min_plane_dis) {
                plane_valid = false;
                break;
            }
        }
        if (plane_valid) {
            bool is_in_laser_fov = check_fov ? check_fov_fn(pose_local, point_sel) : true;
            if (is_in_laser_fov) {
                feature.coeffs[0] = norm[0]; feature.coeffs[1] = norm[1]; feature.coeffs[2] = norm[2];
                feature.coeffs[3] = negative_OA_dot_norm;
                feature.coeffs[4] = feature.coeffs[5] = 0.0;
                feature.idx = idx;
                feature.point[0] = point_ori[0]; feature.point[1] = point_ori[1]; feature.point[2] = point_ori[2];
                feature.laser_idx = (size_t)point_ori[3];
                feature.type = 's';
                return true;
            }
        }
    }
    return false;
}

void match_corner_from_map(const MapCloud &map, const float *cloud_data, size_t stride, int n, const Pose &pose_local,
                           std::vector<Feature> &features, int n_neigh, bool check_fov, const MatchParams &mp)
{
    features.clear();
    for (int i = 0; i < n; ++i) {
        Feature f;
        if (match_corner_point_from_map(map, cloud_data + size_t(i) * stride, pose_local, f, i, n_neigh, check_fov, mp))
            features.push_back(f);
    }
}

void match_surf_from_map(const MapCloud &map, const float *cloud_data, size_t stride, int n, const Pose &pose_local,
                         std::vector<Feature> &features, int n_neigh, bool check_fov, const MatchParams &mp)
{
    features.clear();
    for (int i = 0; i < n; ++i) {
        Feature f;
        if (match_surf_point_from_map(map, cloud_data + size_t(i) * stride, pose_local, f, i, n_neigh, check_fov, mp))
            features.push_back(f);
    }
}

}  // namespace orc
