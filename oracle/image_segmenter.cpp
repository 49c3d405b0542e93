// TEST INFRASTRUCTURE ONLY -- see image_segmenter.hpp.
#include "image_segmenter.hpp"
#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdint>

namespace orc {
namespace {
struct Pt { float x, y, z, intensity; };

struct Setup {   // ImageSegmenter::setParameter (image_segmenter.cpp:18-61)
    int ground_scan_id = 0;
    float ang_res_x = 0, ang_res_y = 0, ang_bottom = 0, alphax = 0, alphay = 0;
};
Setup make_setup(const SegParams &p)
{
    Setup s;
    if (p.vertical_scans == 16) {
        s.ang_res_x = 360.0 / p.horizon_scans; s.ang_res_y = 2.0; s.ang_bottom = 15.0 + 0.1; s.ground_scan_id = 7;
        s.alphax = s.ang_res_x / 180.0 * M_PI; s.alphay = s.ang_res_y / 180.0 * M_PI;
    } else if (p.vertical_scans == 32) {
        s.ang_res_x = 360.0 / p.horizon_scans; s.ang_res_y = 41.33 / float(p.vertical_scans - 1); s.ang_bottom = 30.0 + 0.67; s.ground_scan_id = 20;
        s.alphax = s.ang_res_x / 180.0 * M_PI; s.alphay = s.ang_res_y / 180.0 * M_PI;
    } else if (p.vertical_scans == 64) {
        s.ang_res_x = 360.0 / p.horizon_scans; s.ang_res_y = FLT_MAX; s.ground_scan_id = 63;
        s.alphax = s.ang_res_x / 180.0 * M_PI; s.alphay = 0.f;   // (U3)
    }
    return s;
}
}  // namespace

void segment_cloud(const float *xyzi, int n, const SegParams &prm, SegResult &out)
{
    const int vs = prm.vertical_scans, hs = prm.horizon_scans;
    Setup S = make_setup(prm);
    out = SegResult();
    out.range_mat.assign(size_t(vs) * hs, FLT_MAX);
    out.label_mat.assign(size_t(vs) * hs, 0);
    out.pixel_of_point.assign(size_t(n), -1);
    std::vector<float> &range_mat = out.range_mat;
    std::vector<int> &label_mat = out.label_mat;
    std::vector<Pt> cloud_matrix(size_t(vs) * hs);
    std::vector<std::vector<Pt>> cloud_scan(vs);
    std::vector<int> cloud_scan_order(size_t(vs) * hs, 0);
    auto R = [&](int i, int j) -> float & { return range_mat[size_t(i) * hs + j]; };
    auto L = [&](int i, int j) -> int & { return label_mat[size_t(i) * hs + j]; };

    // ---- projectCloud (hpp:88-136)
    for (int i = 0; i < n; ++i) {
        Pt point{xyzi[4 * i], xyzi[4 * i + 1], xyzi[4 * i + 2], xyzi[4 * i + 3]};
        float range = std::sqrt(point.x * point.x + point.y * point.y + point.z * point.z);
        if (range < prm.roi_range) continue;
        float vertical_angle = std::atan(point.z / std::sqrt(point.x * point.x + point.y * point.y)) * 180 / M_PI;
        int row_id, column_id;
        if ((vs == 64) && (S.ang_res_y == FLT_MAX)) {
            if (vertical_angle >= -8.83) row_id = static_cast<int>((2 - vertical_angle) * 3.0 + 0.5);
            else row_id = static_cast<int>(vs / 2) + static_cast<int>((-8.83 - vertical_angle) * 2.0 + 0.5);
            if (vertical_angle > 2 || vertical_angle < -24.33 || row_id > 50 || row_id < 0) continue;
        } else {
            row_id = static_cast<int>((vertical_angle + S.ang_bottom) / S.ang_res_y);
            if (row_id < 0 || row_id >= vs) continue;
        }
        float horizon_angle = std::atan2(point.x, point.y) * 180 / M_PI;
        column_id = -std::round((horizon_angle - 90.0) / S.ang_res_x) + hs / 2;
        if (column_id >= hs) column_id -= hs;
        if (column_id < 0 || column_id >= hs) continue;
        if (R(row_id, column_id) != FLT_MAX) continue;
        point.intensity += row_id;
        const int index = column_id + row_id * hs;
        cloud_matrix[index] = point;
        R(row_id, column_id) = range;
        cloud_scan[row_id].push_back(point);
        cloud_scan_order[index] = int(cloud_scan[row_id].size()) - 1;
        out.pixel_of_point[i] = index;
    }

    // ---- segmentCloud (hpp:139-393)
    const size_t npx = size_t(vs) * hs;
    std::vector<uint16_t> all_pushed_indx(npx), all_pushed_indy(npx), queue_indx(npx), queue_indy(npx);
    std::vector<int> queue_indx_last_negi(npx, 0), queue_indy_last_negi(npx, 0);
    std::vector<float> queue_last_dis(npx, 0.f);
    for (int i = 0; i < vs; i++)
        for (int j = 0; j < hs; j++)
            if (R(i, j) == FLT_MAX) L(i, j) = -1;

    int label_count = 1;
    {   // ground (hpp:179-227): rows [ground_scan_id, vs) for 64 rings, [0, ground_scan_id) otherwise; (U3): row i + 1 must exist
        const int lo = (vs == 64) ? S.ground_scan_id : 0, hi = (vs == 64) ? vs : S.ground_scan_id;
        for (int i = lo; i < hi && i + 1 < vs; i++) {
            for (int j = 0; j < hs; j++) {
                if (R(i, j) == FLT_MAX || R(i + 1, j) == FLT_MAX) continue;
                const Pt &point1 = cloud_matrix[size_t(j) + size_t(i) * hs];
                const Pt &point2 = cloud_matrix[size_t(j) + size_t(i + 1) * hs];
                float diff_x = point1.x - point2.x, diff_y = point1.y - point2.y, diff_z = point1.z - point2.z;
                float vertical_angle = std::atan2(diff_z, std::sqrt(diff_x * diff_x + diff_y * diff_y)) * 180 / M_PI;
                if (std::abs(vertical_angle) <= 10) { L(i, j) = label_count; L(i + 1, j) = label_count; }
            }
        }
    }
    label_count++;

    static const int8_t nb[4][2] = {{-1, 0}, {0, 1}, {0, -1}, {1, 0}};     // neighbor_iterator_ (hpp:42-54)
    float alpha = 0.f;                                                     // (U1)
    std::vector<char> line_count_flag(vs);
    for (int i = 0; i < vs; i++) {
        for (int j = 0; j < hs; j++) {
            if (L(i, j) != 0) continue;
            float d1, d2, angle, dist;
            std::fill(line_count_flag.begin(), line_count_flag.end(), 0);
            queue_indx[0] = i; queue_indy[0] = j;
            queue_indx_last_negi[0] = 0; queue_indy_last_negi[0] = 0; queue_last_dis[0] = 0;
            int queue_size = 1, queue_start_ind = 0, queue_end_ind = 1;
            all_pushed_indx[0] = i; all_pushed_indy[0] = j;
            int all_pushed_ind_size = 1;
            while (queue_size > 0) {
                const int from_indx = queue_indx[queue_start_ind], from_indy = queue_indy[queue_start_ind];
                --queue_size;
                ++queue_start_ind;
                L(from_indx, from_indy) = label_count;
                for (int q = 0; q < 4; ++q) {
                    int this_indx = from_indx + nb[q][0], this_indy = from_indy + nb[q][1];
                    if (this_indx < 0 || this_indx >= vs) continue;
                    if ((vs == 64) && (S.ang_res_y == FLT_MAX)) {
                        if (this_indx <= 32) S.alphay = 0.333 / 180.0 * M_PI;
                        else S.alphay = 0.5 / 180.0 * M_PI;
                    }
                    if (this_indy < 0) this_indy = hs - 1;
                    if (this_indy >= hs) this_indy = 0;
                    if (L(this_indx, this_indy) != 0) continue;
                    d1 = std::max(R(from_indx, from_indy), R(this_indx, this_indy));
                    d2 = std::min(R(from_indx, from_indy), R(this_indx, this_indy));
                    dist = std::sqrt(d1 * d1 + d2 * d2 - 2 * d1 * d2 * std::cos(alpha));       // alpha of the PREVIOUS neighbour (U1)
                    alpha = nb[q][0] == 0 ? S.alphax : S.alphay;
                    angle = std::atan2(d2 * std::sin(alpha), (d1 - d2 * std::cos(alpha)));
                    bool push = false;
                    if (angle > prm.segment_theta) push = true;
                    else if ((nb[q][1] == 0) && (queue_indy_last_negi[queue_start_ind] == 0)) {
                        const float dist_last = queue_last_dis[queue_start_ind];
                        if ((dist_last / dist <= 1.2) && ((dist_last / dist >= 0.8))) push = true;
                    }
                    if (push) {
                        queue_indx[queue_end_ind] = this_indx; queue_indy[queue_end_ind] = this_indy;
                        queue_indx_last_negi[queue_end_ind] = nb[q][0]; queue_indy_last_negi[queue_end_ind] = nb[q][1];
                        queue_last_dis[queue_end_ind] = dist;
                        queue_size++; queue_end_ind++;
                        L(this_indx, this_indy) = label_count;
                        line_count_flag[this_indx] = 1;
                        all_pushed_indx[all_pushed_ind_size] = this_indx; all_pushed_indy[all_pushed_ind_size] = this_indy;
                        all_pushed_ind_size++;
                    }
                }
            }
            bool feasible_segment = false;
            if (all_pushed_ind_size >= prm.min_cluster_size) feasible_segment = true;
            else if (all_pushed_ind_size >= prm.segment_valid_point_num) {
                int line_count = 0;
                for (int r = 0; r < vs; r++) if (line_count_flag[r]) line_count++;
                if (line_count >= prm.segment_valid_line_num) feasible_segment = true;
            }
            if (feasible_segment) label_count++;
            else for (int k = 0; k < all_pushed_ind_size; ++k) L(all_pushed_indx[k], all_pushed_indy[k]) = 999999;
        }
    }

    // ---- outliers out, rows concatenated (hpp:362-392)
    std::vector<Pt> outlier;
    if (prm.segment_flag) {
        for (int i = 0; i < vs; i++)
            for (int j = 0; j < hs; j++)
                if (L(i, j) > 0 && L(i, j) == 999999) {
                    const int index = j + i * hs;
                    const int pos = cloud_scan_order[index];
                    if (pos >= 0 && size_t(pos) < cloud_scan[i].size()) cloud_scan[i].erase(cloud_scan[i].begin() + pos);     // (U2)
                    if (j % 5 == 0) outlier.push_back(cloud_matrix[index]);
                }
    }
    out.scan_start.resize(vs); out.scan_end.resize(vs);
    std::vector<Pt> cloud_out;
    for (int i = 0; i < vs; i++) {
        out.scan_start[i] = int(cloud_out.size()) + 5;
        cloud_out.insert(cloud_out.end(), cloud_scan[i].begin(), cloud_scan[i].end());
        out.scan_end[i] = int(cloud_out.size()) - 6;
    }
    if (!cloud_out.empty()) outlier.push_back(cloud_out[0]);               // hpp:391 (UB on an empty cloud; nothing is pushed here)
    out.cloud_out.resize(cloud_out.size() * 4);
    for (size_t k = 0; k < cloud_out.size(); ++k) { out.cloud_out[4 * k] = cloud_out[k].x; out.cloud_out[4 * k + 1] = cloud_out[k].y; out.cloud_out[4 * k + 2] = cloud_out[k].z; out.cloud_out[4 * k + 3] = cloud_out[k].intensity; }
    out.cloud_outlier.resize(outlier.size() * 4);
    for (size_t k = 0; k < outlier.size(); ++k) { out.cloud_outlier[4 * k] = outlier[k].x; out.cloud_outlier[4 * k + 1] = outlier[k].y; out.cloud_outlier[4 * k + 2] = outlier[k].z; out.cloud_outlier[4 * k + 3] = outlier[k].intensity; }
}

}  // namespace orc
