// Exercises the C++ facade end to end on binary inputs written by the python test harness (tests/test_gpu_facade.py):
//   <dir>/scan.f32 (n x 4), <dir>/rings.i32 (2 x n_rings: starts then ends), <dir>/surf_map.f32, <dir>/corner_map.f32 (n x 3),
//   <dir>/surf.f32, <dir>/corner.f32 (m x 4: x y z lidar-id), <dir>/pose.f64 (7)
// and writes <dir>/out_labels.i32, <dir>/out_pose.f64 (7), <dir>/out_counts.i32 (per outer: n_surf, n_corner, lm_iterations),
// <dir>/out_valid_surf.u8 (batch matchSurfFromMap at the initial pose).
#include "mloam_facade.hpp"
#include <random>
#include <cmath>
#include <cstdio>
#include <fstream>
#include <omp.h>

using namespace mloam_hip;

template <typename T>
static std::vector<T> read_file(const std::string &path)
{
    std::ifstream f(path, std::ios::binary | std::ios::ate);
    if (!f) throw Error("cannot open " + path);
    size_t bytes = (size_t)f.tellg();
    std::vector<T> v(bytes / sizeof(T));
    f.seekg(0);
    f.read(reinterpret_cast<char *>(v.data()), bytes);
    return v;
}
template <typename T>
static void write_file(const std::string &path, const std::vector<T> &v)
{
    std::ofstream f(path, std::ios::binary);
    f.write(reinterpret_cast<const char *>(v.data()), sizeof(T) * v.size());
}

static PointICovCloud cov_cloud(const std::vector<float> &a, int cols)
{
    PointICovCloud c;
    for (size_t i = 0; i + cols <= a.size(); i += cols) {
        PointIWithCov p;
        p.x = a[i]; p.y = a[i + 1]; p.z = a[i + 2];
        p.intensity = cols > 3 ? a[i + 3] : 0.f;
        c.push_back(p);
    }
    return c;
}

int main(int argc, char **argv)
{
    std::setvbuf(stdout, nullptr, _IONBF, 0);   // a crash must not take the progress lines with it
    if (argc < 2) { std::fprintf(stderr, "usage: %s <dir>\n", argv[0]); return 2; }
    const std::string d = std::string(argv[1]) + "/";
    try {
        Device dev(0);
        // --- extractCloud
        auto scan = read_file<float>(d + "scan.f32");
        auto rings = read_file<int>(d + "rings.i32");
        const int n_rings = (int)rings.size() / 2;
        PointICloud cloud;
        for (size_t i = 0; i + 4 <= scan.size(); i += 4) { PointI p; p.x = scan[i]; p.y = scan[i + 1]; p.z = scan[i + 2]; p.intensity = scan[i + 3]; cloud.push_back(p); }
        ScanInfo info(n_rings, false);
        for (int r = 0; r < n_rings; ++r) { info.scan_start_ind_[r] = rings[r]; info.scan_end_ind_[r] = rings[n_rings + r]; }
        FeatureExtract f_extract(dev);
        cloudFeature cf;
        f_extract.extractCloud(cloud, info, cf);
        write_file(d + "out_labels.i32", f_extract.cloudLabel());
        {
            std::vector<float> lf;
            for (const auto &q : cf["surf_points_less_flat"].points) { lf.push_back(q.x); lf.push_back(q.y); lf.push_back(q.z); lf.push_back(q.intensity); }
            write_file(d + "out_less_flat.f32", lf);
        }
        std::printf("extract: sharp %zu less_sharp %zu flat %zu less_flat %zu\n", cf["corner_points_sharp"].size(),
                    cf["corner_points_less_sharp"].size(), cf["surf_points_flat"].size(), cf["surf_points_less_flat"].size());
        // --- batch matching through the FeatureExtract signature
        PointICovCloud surf_map = cov_cloud(read_file<float>(d + "surf_map.f32"), 3), corner_map = cov_cloud(read_file<float>(d + "corner_map.f32"), 3);
        PointICovCloud surf = cov_cloud(read_file<float>(d + "surf.f32"), 4), corner = cov_cloud(read_file<float>(d + "corner.f32"), 4);
        auto pv = read_file<double>(d + "pose.f64");
        Pose pose;
        pose.fromParam(pv.data());
        MapIndex<PointIWithCov> kd_surf(dev, MLH_SURF);
        kd_surf.setInputCloud(surf_map);
        std::vector<PointPlaneFeature> feats;
        f_extract.matchSurfFromMap(kd_surf, surf_map, surf, pose, feats, 5, false);
        std::vector<uint8_t> valid(surf.size(), 0);
        for (const auto &f : feats) valid[f.idx_] = 1;
        write_file(d + "out_valid_surf.u8", valid);
        std::printf("matchSurfFromMap: %zu of %zu\n", feats.size(), surf.size());
        // --- scan2MapOptimization
        Scan2MapReport rep;
        scan2MapOptimization(dev, surf_map, corner_map, surf, corner, pose, false, &rep);
        double out[7];
        pose.toParam(out);
        write_file(d + "out_pose.f64", std::vector<double>(out, out + 7));
        std::vector<int> counts;
        for (const auto &s : rep.outer) { counts.push_back(s.n_surf); counts.push_back(s.n_corner); counts.push_back(s.lm_iterations); }
        write_file(d + "out_counts.i32", counts);
        std::printf("scan2map pose: %.9f %.9f %.9f  %.9f %.9f %.9f %.9f\n", out[0], out[1], out[2], out[3], out[4], out[5], out[6]);
        // --- VoxelGridCovarianceMLOAM through the reference's setter names
        VoxelGridCovarianceMLOAM<PointIWithCov> down_size_filter_surf_map_cov(dev);
        down_size_filter_surf_map_cov.setLeafSize(0.8f, 0.8f, 0.8f);
        down_size_filter_surf_map_cov.setTraceThreshold(1.0f);
        down_size_filter_surf_map_cov.setInputCloud(surf_map);
        PointICovCloud surf_map_ds;
        down_size_filter_surf_map_cov.filter(surf_map_ds);
        std::vector<float> ds;
        for (const auto &q : surf_map_ds.points) { ds.push_back(q.x); ds.push_back(q.y); ds.push_back(q.z); ds.push_back(q.intensity); for (int k = 0; k < 6; ++k) ds.push_back(q.cov_vec[k]); ds.push_back(q.cov_trace); }
        write_file(d + "out_map_ds.f32", ds);
        std::printf("voxel filter: %zu -> %zu\n", surf_map.size(), surf_map_ds.size());
        // --- cloudUCTAssociateToMap with the reference's argument list
        {
            std::vector<Pose> pose_ext(2);
            pose_ext[1].t_(0) = 0.1; pose_ext[1].t_(1) = -0.5; pose_ext[1].q_.z = 0.0998334166; pose_ext[1].q_.w = 0.9950041653;
            for (int i = 0; i < 6; ++i) pose_ext[1].cov_[i * 6 + i] = i < 3 ? 0.0025 : 0.00030461;
            Pose pose_global = pose;
            for (int i = 0; i < 6; ++i) pose_global.cov_[i * 6 + i] = 1e-5;
            PointICovCloud kf, kf_map;
            for (size_t i = 0; i < surf.size(); ++i) { PointIWithCov q = surf[i]; q.intensity = float(i & 1); kf.push_back(q); }
            cloudUCTAssociateToMap(dev, kf, kf_map, pose_global, pose_ext, true);
            std::vector<float> km;
            for (const auto &q : kf_map.points) { km.push_back(q.x); km.push_back(q.y); km.push_back(q.z); km.push_back(q.intensity); for (int k = 0; k < 6; ++k) km.push_back(q.cov_vec[k]); km.push_back(q.cov_trace); }
            write_file(d + "out_kf_map.f32", km);
            std::printf("cloudUCTAssociateToMap: %zu -> %zu\n", kf.size(), kf_map.size());
        }
        // --- LidarPureOdomBatchFactor: the matched surf features as window factors over [pivot, 2 frames, 2 extrinsics]
        {
            LidarPureOdomBatchFactor cost(dev, 2, 2);
            for (size_t i = 0; i < feats.size(); ++i) cost.add(feats[i], 1 + int(i % 2), int((i / 2) % 2));
            double pivot[7] = {0.3, -0.2, 0.1, 0.0, 0.0, 0.0499791693, 0.9987502604}, f1[7], f2[7];
            pose.toParam(f1); pose.toParam(f2); f2[0] += 0.25;
            double e0[7] = {0, 0, 0, 0, 0, 0, 1}, e1[7] = {0.1, -0.5, 0.02, 0.0, 0.0, 0.0998334166, 0.9950041653};
            const double *par[5] = {pivot, f1, f2, e0, e1};
            const int nr = cost.num_residuals();
            std::vector<double> res(nr), jac(size_t(nr) * 7 * 5);
            double *jp[5];
            for (int b = 0; b < 5; ++b) jp[b] = jac.data() + size_t(b) * nr * 7;
            if (!cost.Evaluate(par, res.data(), jp)) throw Error("LidarPureOdomBatchFactor::Evaluate");
            write_file(d + "out_odom_res.f64", res);
            write_file(d + "out_odom_jac.f64", jac);
            std::printf("LidarPureOdomBatchFactor: %d residuals\n", nr);
        }
        // --- LidarTracker::trackCloud on the two tracker scans (files written by the test: [x y z ring] rows)
        {
            auto load = [&](const char *name) { PointICloud c; auto a = read_file<float>(d + name); for (size_t i = 0; i + 4 <= a.size(); i += 4) { PointI p; p.x = a[i]; p.y = a[i + 1]; p.z = a[i + 2]; p.intensity = a[i + 3]; c.push_back(p); } return c; };
            cloudFeature prev_f, cur_f;
            prev_f["corner_points_less_sharp"] = load("trk_corner_last.f32"); prev_f["surf_points_less_flat"] = load("trk_surf_last.f32");
            cur_f["corner_points_sharp"] = load("trk_corner_sharp.f32"); cur_f["surf_points_flat"] = load("trk_surf_flat.f32");
            LidarTracker tracker(dev);
            const Pose pose_prev_cur = tracker.trackCloud(prev_f, cur_f, Pose());
            double tp[7];
            pose_prev_cur.toParam(tp);
            write_file(d + "out_track_pose.f64", std::vector<double>(tp, tp + 7));
            std::printf("trackCloud: %.6f %.6f %.6f\n", tp[0], tp[1], tp[2]);
        }
        // --- the same front end without host hops: extractCloudOnDevice -> LidarTracker / fuseCloudFeature -> downsampleFusedScan
        {
            auto load_scan = [&](const char *pts, const char *rg, PointICloud &c, ScanInfo &si) {
                auto a = read_file<float>(d + pts);
                auto r = read_file<int>(d + rg);
                const int nr = (int)r.size() / 2;
                for (size_t i = 0; i + 4 <= a.size(); i += 4) { PointI p; p.x = a[i]; p.y = a[i + 1]; p.z = a[i + 2]; p.intensity = a[i + 3]; c.push_back(p); }
                si = ScanInfo(nr, false);
                for (int k = 0; k < nr; ++k) { si.scan_start_ind_[k] = r[k]; si.scan_end_ind_[k] = r[nr + k]; }
            };
            PointICloud c0, c1;
            ScanInfo i0(1, false), i1(1, false);
            load_scan("trk_scan_prev.f32", "trk_rings_prev.i32", c0, i0);
            load_scan("trk_scan_cur.f32", "trk_rings_cur.i32", c1, i1);
            LidarTracker tracker(dev);
            f_extract.extractCloudOnDevice(c0, i0);
            tracker.setPrevFromExtractor();
            f_extract.extractCloudOnDevice(c1, i1);
            tracker.setCurFromExtractor();
            double tp[7];
            tracker.trackCloudOnDevice(Pose()).toParam(tp);
            write_file(d + "out_track_pose_dev.f64", std::vector<double>(tp, tp + 7));
            std::vector<Pose> pose_ext(2);
            pose_ext[1].t_(0) = 0.1; pose_ext[1].t_(1) = -0.5; pose_ext[1].q_.z = 0.0998334166; pose_ext[1].q_.w = 0.9950041653;
            pose_ext[1].cov_[0] = pose_ext[1].cov_[7] = pose_ext[1].cov_[14] = 0.0025;
            pose_ext[1].cov_[21] = pose_ext[1].cov_[28] = pose_ext[1].cov_[35] = 0.00030461;
            fuseReset(dev);
            fuseCloudFeature(dev, 1, pose_ext[1]);            // the scan still held by the device, as LiDAR 1
            f_extract.extractCloudOnDevice(c0, i0);
            fuseCloudFeature(dev, 0, pose_ext[0]);
            std::vector<int32_t> kept = {downsampleFusedScan(dev, MLH_SURF, 0.4f, pose_ext, true), downsampleFusedScan(dev, MLH_CORNER, 0.2f, pose_ext, true)};
            const std::pair<int, int> both = downsampleFusedScans(dev, 0.4f, 0.2f, pose_ext, true);     // the same two clouds through one pipeline
            kept.push_back(both.first); kept.push_back(both.second);
            write_file(d + "out_fused_kept.i32", kept);
            std::printf("device-resident front end: track %.6f %.6f %.6f, fused features %d + %d\n", tp[0], tp[1], tp[2], kept[0], kept[1]);
            // the odometry's window map: transformPointCloud + pcl::VoxelGrid
            {
                PointICloud moved, thin;
                transformPointCloud(dev, c0, moved, pose_ext[1]);
                VoxelGrid vg(dev);
                vg.setLeafSize(0.3f, 0.3f, 0.3f);
                vg.setInputCloud(moved);
                vg.filter(thin);
                std::vector<float> to;
                for (const auto &q : thin.points) { to.push_back(q.x); to.push_back(q.y); to.push_back(q.z); to.push_back(q.intensity); }
                write_file(d + "out_window_map.f32", to);
            }
            // undistortion (DISTORTION = 1): TransformToEnd over a host cloud; the device-resident variant on the scan just extracted
            PointICloud und = c1;
            Pose pose_undist;
            pose_undist.t_(0) = 0.35; pose_undist.t_(1) = -0.12; pose_undist.t_(2) = 0.02; pose_undist.q_.z = 0.0130895956; pose_undist.q_.w = 0.9999143276;
            TransformToEnd(dev, und, pose_undist, true, 0.1f);
            std::vector<float> uo;
            for (const auto &q : und.points) { uo.push_back(q.x); uo.push_back(q.y); uo.push_back(q.z); uo.push_back(q.intensity); }
            write_file(d + "out_undistorted.f32", uo);
            undistortMeasurementsOnDevice(dev, pose_undist, 0.1f);
        }
        // --- the per-feature Ceres contract + ActiveFeatureSelection + ImageSegmenter (round 2)
        {
            // LidarMap{PlaneNorm,Edge}Factor: (point, coeff, cov) -> Evaluate(parameters, residuals, jacobians), also with null Jacobians
            auto surf_f = read_file<float>(d + "surf.f32");
            auto corner_f = read_file<float>(d + "corner.f32");
            auto surf_m = read_file<float>(d + "surf_map.f32");
            auto corner_m = read_file<float>(d + "corner_map.f32");
            auto pose_in = read_file<double>(d + "pose.f64");
            Pose pose0;
            pose0.fromParam(pose_in.data());
            PointICovCloud map_s = cov_cloud(surf_m, 3), map_c = cov_cloud(corner_m, 3), feat_s = cov_cloud(surf_f, 4), feat_c = cov_cloud(corner_f, 4);
            for (auto *c : {&feat_s, &feat_c}) for (auto &q : c->points) { q.cov_vec[0] = 0.01f; q.cov_vec[3] = 0.02f; q.cov_vec[5] = 0.03f; }
            MapIndex<PointIWithCov> kd_s(dev, MLH_SURF), kd_c(dev, MLH_CORNER);
            kd_s.setInputCloud(map_s);
            kd_c.setInputCloud(map_c);
            ActiveFeatureSelection afs(dev, true, 11);
            std::array<double, 36> mat_H{};
            for (int i = 0; i < 6; ++i) mat_H[i * 7] = 1e-6;
            int total_feat_num = 0;
            afs.evalFullHessian(kd_s, map_s, feat_s, pose0, 's', mat_H, total_feat_num);
            afs.evalFullHessian(kd_c, map_c, feat_c, pose0, 'c', mat_H, total_feat_num);
            const double gf_deg_factor = logDet6(mat_H.data());
            std::vector<double> afs_out(mat_H.begin(), mat_H.end());
            afs_out.push_back(double(total_feat_num));
            afs_out.push_back(gf_deg_factor);
            for (const char *m : {"wo_gf", "rnd", "fps", "gd_fix", "gd_float"})
                for (double thre : {gf_deg_factor - 1.0, gf_deg_factor + 1.0}) afs_out.push_back(gfRatioPolicy(m, 0.2, gf_deg_factor, thre));
            write_file(d + "out_afs.f64", afs_out);
            // goodFeatureMatching + the reference's block-assembly loop (lidar_mapper_keyframe.cpp:537-571) on the selected features
            std::vector<PointPlaneFeature> all_surf_features;
            std::vector<size_t> sel_surf_feature_idx;
            std::array<double, 36> sub_mat_H{};
            for (int i = 0; i < 6; ++i) sub_mat_H[i * 7] = 1e-6;
            afs.goodFeatureMatching(kd_s, map_s, feat_s, pose0, all_surf_features, sel_surf_feature_idx, 's', "gd_fix", 0.2, sub_mat_H);
            std::vector<double> rows;          // per selected feature: idx, residual, 7 Jacobian entries
            double para_pose[7];
            pose0.toParam(para_pose);
            for (const size_t &fid : sel_surf_feature_idx) {
                const PointPlaneFeature &feature = all_surf_features[fid];
                std::array<double, 9> cov_matrix{};
                const float *cv = feat_s.points[feature.idx_].cov_vec;       // extractCov (point_with_cov.hpp:202)
                cov_matrix = {cv[0], cv[1], cv[2], cv[1], cv[3], cv[4], cv[2], cv[4], cv[5]};
                LidarMapPlaneNormFactor *f = new LidarMapPlaneNormFactor(feature.point_, feature.coeffs_, cov_matrix);
                double res, jac[7];
                const double *params1[1] = {para_pose};
                double *jacs[1] = {jac};
                f->Evaluate(params1, &res, jacs);
                double res2;
                f->Evaluate(params1, &res2, nullptr);
                if (res2 != res) throw Error("null-Jacobian Evaluate disagrees");
                rows.push_back(double(fid)); rows.push_back(res);
                for (int k = 0; k < 7; ++k) rows.push_back(jac[k]);
                delete f;
            }
            write_file(d + "out_sel_rows.f64", rows);
            std::vector<double> subh(sub_mat_H.begin(), sub_mat_H.end());
            write_file(d + "out_sel_H.f64", subh);
            // one edge factor per matched corner feature through the same contract
            std::vector<PointPlaneFeature> all_corner_features;
            std::vector<size_t> sel_corner_feature_idx;
            std::array<double, 36> sub_c{};
            for (int i = 0; i < 6; ++i) sub_c[i * 7] = 1e-6;
            afs.goodFeatureMatching(kd_c, map_c, feat_c, pose0, all_corner_features, sel_corner_feature_idx, 'c', "wo_gf", 1.0, sub_c);
            std::vector<double> crow;
            for (const size_t &fid : sel_corner_feature_idx) {
                const PointPlaneFeature &feature = all_corner_features[fid];
                LidarMapEdgeFactor f(feature.point_, feature.coeffs_, std::array<double, 9>{0.0025, 0, 0, 0, 0.0025, 0, 0, 0, 0.0025});
                double res, jac[7];
                const double *params1[1] = {para_pose};
                double *jacs[1] = {jac};
                f.Evaluate(params1, &res, jacs);
                crow.push_back(double(fid)); crow.push_back(res);
                for (int k = 0; k < 7; ++k) crow.push_back(jac[k]);
            }
            write_file(d + "out_corner_rows.f64", crow);
            // ImageSegmenter: the unordered cloud written by the harness -> ring-major cloud + ScanInfo -> extractCloud on what it staged
            auto raw = read_file<float>(d + "raw_cloud.f32");
            PointICloud raw_cloud, seg_out, seg_outlier;
            for (size_t i = 0; i + 4 <= raw.size(); i += 4) { PointI q; q.x = raw[i]; q.y = raw[i + 1]; q.z = raw[i + 2]; q.intensity = raw[i + 3]; raw_cloud.push_back(q); }
            ImageSegmenter img_segment(dev);
            img_segment.setParameter(16, 1800, 30, 5, 3);
            ScanInfo seg_info(16, true);
            img_segment.segmentCloud(raw_cloud, seg_out, seg_outlier, seg_info);
            std::vector<float> so;
            for (const auto &q : seg_out.points) { so.push_back(q.x); so.push_back(q.y); so.push_back(q.z); so.push_back(q.intensity); }
            write_file(d + "out_seg_cloud.f32", so);
            std::vector<int> sinfo(seg_info.scan_start_ind_);
            sinfo.insert(sinfo.end(), seg_info.scan_end_ind_.begin(), seg_info.scan_end_ind_.end());
            write_file(d + "out_seg_info.i32", sinfo);
            // the window's factor table built on the device from two match passes (surf: frame 1 / LiDAR 0, corner: frame 1 / LiDAR 1), then the
            // coupled normal equations over [pivot | 1 frame | 2 extrinsics] (D = 24, the hercules window)
            {
                PointICloud fs_i, fc_i;
                for (size_t i = 0; i + 4 <= surf_f.size(); i += 4) { PointI q; q.x = surf_f[i]; q.y = surf_f[i + 1]; q.z = surf_f[i + 2]; q.intensity = surf_f[i + 3]; fs_i.push_back(q); }
                for (size_t i = 0; i + 4 <= corner_f.size(); i += 4) { PointI q; q.x = corner_f[i]; q.y = corner_f[i + 1]; q.z = corner_f[i + 2]; q.intensity = corner_f[i + 3]; fc_i.push_back(q); }
                WindowFactorTable table(dev);
                table.addMatches(fs_i, 's', pose0, 1, 0, 5, true);
                table.addMatches(fc_i, 'c', pose0, 1, 1, 10, true);
                std::array<double, 7> piv = {0, 0, 0, 0, 0, 0, 1}, fr{}, e0 = piv, e1 = piv;
                pose0.toParam(fr.data());
                WindowNormalEquations wne;
                evalWindowNormalEquations(dev, piv.data(), {fr}, {e0, e1}, 1.0, wne);
                std::vector<double> wout(wne.JtJ);
                wout.insert(wout.end(), wne.Jtr.begin(), wne.Jtr.end());
                wout.push_back(wne.cost); wout.push_back(double(wne.n_residuals));
                write_file(d + "out_window_ne.f64", wout);
                // the odometry's own selection in front of the table (Estimator::goodFeatureMatching, estimator.cpp:1347-1517): pivot = identity, pose_i = pose0,
                // the second LiDAR's extrinsic = identity; ODOM_GF_RATIO 0.8 for the surf group, 0.3 for the corner group
                {
                    WindowFactorTable sel_table(dev);
                    Pose ident_pose;
                    std::vector<size_t> sel_s, sel_c;
                    sel_table.goodFeatureMatching(fs_i, sel_s, 's', ident_pose, pose0, ident_pose, 0.8f, 1, 0, 11);
                    sel_table.goodFeatureMatching(fc_i, sel_c, 'c', ident_pose, pose0, ident_pose, 0.3f, 1, 1, 12);
                    WindowNormalEquations sne;
                    evalWindowNormalEquations(dev, piv.data(), {fr}, {e0, e1}, 1.0, sne);
                    std::vector<int> sel_out;
                    sel_out.push_back(int(sel_s.size())); sel_out.push_back(int(sel_c.size())); sel_out.push_back(int(sne.n_residuals));
                    for (size_t q : sel_s) sel_out.push_back(int(q));
                    for (size_t q : sel_c) sel_out.push_back(int(q));
                    write_file(d + "out_odom_selection.i32", sel_out);
                }
                setVoxelMemberOrderAsReference(dev, false);   // device-only member order ...
                setVoxelMemberOrderAsReference(dev, true);    // ... and back to the default (the reference's)
            }
            std::printf("round-2 facade: full-H features %d, logdet %.6f, selected %zu surf rows, %zu corner rows, segmented %zu -> %zu points\n", total_feat_num,
                        gf_deg_factor, sel_surf_feature_idx.size(), sel_corner_feature_idx.size(), raw_cloud.size(), seg_out.size());
        }
        // --- round 3: FramePipeline (split submission, overlapped map staging, start pose chained on the device) against the synchronous calls
        {
            Pose start;
            start.fromParam(pv.data());
            Pose od0, od1;                                            // two odometry poses a few centimetres apart
            double o0[7] = {0.3, -0.2, 0.1, 0.0, 0.0, 0.001, 1.0}, o1[7] = {0.33, -0.18, 0.1, 0.0, 0.0005, 0.0012, 1.0};
            for (double *o : {o0, o1}) { const double n = std::sqrt(o[3] * o[3] + o[4] * o[4] + o[5] * o[5] + o[6] * o[6]); for (int i = 3; i < 7; ++i) o[i] /= n; }
            od0.fromParam(o0); od1.fromParam(o1);
            FramePipeline pipe(dev, 3);
            pipe.setInputClouds(surf_map, corner_map);                // nothing in flight: the plain staging path
            pipe.setFeatures(surf, corner);
            pipe.submit(start);
            pipe.setInputClouds(surf_map, corner_map);                // frame 1's maps while frame 0 is being solved
            pipe.submitChained(od0, od1);
            double pa[7], pb[7];
            pipe.collect().toParam(pa);
            pipe.collect().toParam(pb);
            std::vector<double> po(pa, pa + 7);
            po.insert(po.end(), pb, pb + 7);
            po.insert(po.end(), o0, o0 + 7);
            po.insert(po.end(), o1, o1 + 7);
            write_file(d + "out_pipeline.f64", po);
            std::printf("frame pipeline: %.9f %.9f %.9f -> %.9f %.9f %.9f\n", pa[0], pa[1], pa[2], pb[0], pb[1], pb[2]);
        }
        // --- round 4: the estimator's front end AS THE REFERENCE WRITES IT (estimator.cpp:249-263): ONE ImageSegmenter and ONE FeatureExtract as members,
        //     NUM_OF_LASER OpenMP threads calling calTimestamp / segmentCloud / extractCloud on them at once. Every LiDAR's result must equal what the same
        //     calls give one after the other on an object bound to one context.
        {
            const int NUM_OF_LASER = 4, N_SCANS = 16;
            auto raw = read_file<float>(d + "raw_cloud.f32");
            std::vector<PointXYZCloud> v_laser_cloud_in(NUM_OF_LASER);
            for (int l = 0; l < NUM_OF_LASER; ++l) {             // four different scans: the harness's cloud turned about z by 0 / 17 / 34 / 51 degrees
                const double a = l * 17.0 * M_PI / 180.0, ca = std::cos(a), sa = std::sin(a);
                for (size_t i = 0; i + 4 <= raw.size(); i += 4) {
                    PointXYZ q;
                    q.x = float(ca * raw[i] - sa * raw[i + 1]); q.y = float(sa * raw[i] + ca * raw[i + 1]); q.z = raw[i + 2];
                    v_laser_cloud_in[l].push_back(q);
                }
            }
            auto flatten = [](cloudFeature &cf_, std::vector<float> &o) {
                o.clear();
                for (const char *k : {"laser_cloud", "corner_points_sharp", "corner_points_less_sharp", "surf_points_flat", "surf_points_less_flat", "laser_cloud_outlier"}) {
                    o.push_back(float(cf_[k].size()));
                    for (const auto &q : cf_[k].points) { o.push_back(q.x); o.push_back(q.y); o.push_back(q.z); o.push_back(q.intensity); }
                }
            };
            // one after the other, bound objects
            std::vector<std::vector<float>> seq(NUM_OF_LASER), par(NUM_OF_LASER);
            std::vector<std::vector<int32_t>> seq_labels(NUM_OF_LASER), par_labels(NUM_OF_LASER);
            std::vector<float> stamps;
            {
                ImageSegmenter seg_b(dev);
                seg_b.setParameter(N_SCANS, 1800, 30, 5, 3);
                FeatureExtract fe_b(dev);
                for (int i = 0; i < NUM_OF_LASER; ++i) {
                    PointICloud laser_cloud, laser_cloud_segment, laser_cloud_outlier;
                    fe_b.calTimestamp(v_laser_cloud_in[i], laser_cloud);
                    if (i == 0) for (const auto &q : laser_cloud.points) stamps.push_back(q.intensity);
                    ScanInfo scan_info(N_SCANS, true);
                    seg_b.segmentCloud(laser_cloud, laser_cloud_segment, laser_cloud_outlier, scan_info);
                    cloudFeature cfb;
                    fe_b.extractCloud(laser_cloud_segment, scan_info, cfb);
                    cfb.insert(std::pair<std::string, PointICloud>("laser_cloud_outlier", laser_cloud_outlier));
                    flatten(cfb, seq[i]);
                    seq_labels[i] = fe_b.cloudLabel();
                }
            }
            write_file(d + "out_timestamps.f32", stamps);
            // the reference's loop: members without a context, all LiDARs at once
            ImageSegmenter img_segment_;
            img_segment_.setParameter(N_SCANS, 1800, 30, 5, 3);
            FeatureExtract f_extract_;
            std::vector<cloudFeature *> feature_frame_ptr(NUM_OF_LASER);
            std::vector<int> thread_of(NUM_OF_LASER, -1);
            for (int round = 0; round < 3; ++round) {           // three frames: the per-thread contexts are created once and reused
#pragma omp parallel for num_threads(NUM_OF_LASER)
                for (size_t i = 0; i < v_laser_cloud_in.size(); i++) {
                    PointICloud laser_cloud;
                    f_extract_.calTimestamp(v_laser_cloud_in[i], laser_cloud);
                    PointICloud laser_cloud_segment, laser_cloud_outlier;
                    ScanInfo scan_info(N_SCANS, true);
                    img_segment_.segmentCloud(laser_cloud, laser_cloud_segment, laser_cloud_outlier, scan_info);
                    feature_frame_ptr[i] = new cloudFeature;
                    f_extract_.extractCloud(laser_cloud_segment, scan_info, *feature_frame_ptr[i]);
                    feature_frame_ptr[i]->insert(std::pair<std::string, PointICloud>("laser_cloud_outlier", laser_cloud_outlier));
                    par_labels[i] = f_extract_.cloudLabel();
                    thread_of[i] = omp_get_thread_num();
                }
                for (int i = 0; i < NUM_OF_LASER; ++i) { flatten(*feature_frame_ptr[i], par[i]); delete feature_frame_ptr[i]; }
            }
            std::vector<int> verdict;
            int distinct_threads = 0;
            { std::vector<int> seen(64, 0); for (int t : thread_of) if (t >= 0 && t < 64 && !seen[t]) { seen[t] = 1; ++distinct_threads; } }
            for (int i = 0; i < NUM_OF_LASER; ++i) {
                verdict.push_back(par[i] == seq[i] ? 1 : 0);
                verdict.push_back(par_labels[i] == seq_labels[i] ? 1 : 0);
                verdict.push_back(int(seq_labels[i].size()));
            }
            verdict.push_back(distinct_threads);
            write_file(d + "out_reentrant.i32", verdict);
            // round 6: the same loop from ONE thread through the facade's own lanes (FrontEndLanes::processAllLasers; a caller without OpenMP): three frames on
            // two lanes (four LiDARs -> two passes) and on four, every LiDAR's clouds equal to the one-after-the-other run; a job's exception arrives at the caller
            {
                std::vector<int> lv;
                for (int n_lanes : {2, 4}) {
                    FrontEndLanes lanes(n_lanes);
                    std::vector<cloudFeature> feature_frame;
                    int equal = 0;
                    for (int round = 0; round < 3; ++round) {
                        lanes.processAllLasers(img_segment_, f_extract_, v_laser_cloud_in, N_SCANS, true, feature_frame);
                        equal = 0;
                        for (int i = 0; i < NUM_OF_LASER; ++i) { std::vector<float> o; flatten(feature_frame[size_t(i)], o); equal += o == seq[size_t(i)] ? 1 : 0; }
                    }
                    lv.push_back(equal);
                }
                {
                    FrontEndLanes lanes(1);
                    int caught = 0;
                    lanes.post(0, [] { throw Error("lane job failed"); });
                    try { lanes.wait(0); } catch (const Error &) { caught = 1; }
                    lanes.post(0, [] {});
                    lanes.wait(0);                                  // the lane survives a failed job
                    lv.push_back(caught);
                }
                // FeatureExtract::sendAhead: the cloud sent to the device ahead of the extractCloud that is handed the same object -- the same clouds, and the library
                // says the look-ahead served the call
                {
                    FeatureExtract fe_b(dev);
                    ImageSegmenter seg_b(dev);
                    seg_b.setParameter(N_SCANS, 1800, 30, 5, 3);
                    PointICloud laser_cloud, laser_cloud_segment, laser_cloud_outlier;
                    fe_b.calTimestamp(v_laser_cloud_in[1], laser_cloud);
                    ScanInfo scan_info(N_SCANS, true);
                    seg_b.segmentCloud(laser_cloud, laser_cloud_segment, laser_cloud_outlier, scan_info);
                    cloudFeature plain, ahead;
                    fe_b.extractCloud(laser_cloud_segment, scan_info, plain);
                    mlh_device_info i0, i1;
                    dev.check(mlh_get_info(dev.ctx(), &i0));
                    fe_b.sendAhead(laser_cloud_segment);
                    fe_b.extractCloud(laser_cloud_segment, scan_info, ahead);
                    dev.check(mlh_get_info(dev.ctx(), &i1));
                    std::vector<float> a, b;
                    flatten(plain, a); flatten(ahead, b);
                    lv.push_back(a == b ? 1 : 0);
                    lv.push_back(int(i1.scan_uploads_from_ahead - i0.scan_uploads_from_ahead));
                }
                // the front end without a host hop: segmentCloudOnDevice -> extractStagedCloudOnDevice per LiDAR, then downsampleFusedScans -- on ONE Device, the
                // LiDARs one after the other, against the lanes (a context each, gathered by fuseCloudFeatureFrom; two lanes = two passes, and four), three frames:
                // the thinned feature counts must agree every time
                {
                    std::vector<Pose> pose_ext(NUM_OF_LASER);
                    for (int i = 0; i < NUM_OF_LASER; ++i) { pose_ext[size_t(i)].t_(0) = 0.1 * i; pose_ext[size_t(i)].t_(1) = -0.05 * i; }
                    ImageSegmenter seg_b(dev);
                    seg_b.setParameter(N_SCANS, 1800, 30, 5, 3);
                    FeatureExtract fe_b(dev);
                    fuseReset(dev);
                    for (int i = 0; i < NUM_OF_LASER; ++i) {
                        PointICloud laser_cloud;
                        fe_b.calTimestamp(v_laser_cloud_in[size_t(i)], laser_cloud);
                        ScanInfo scan_info(N_SCANS, true);
                        seg_b.segmentCloudOnDevice(laser_cloud, scan_info);
                        fe_b.extractStagedCloudOnDevice();
                        fuseCloudFeature(dev, i, pose_ext[size_t(i)]);
                    }
                    const std::pair<int, int> want = downsampleFusedScans(dev, 0.4f, 0.2f, pose_ext, true);
                    int agree = 0, tried = 0;
                    for (int n_lanes : {2, 4}) {
                        FrontEndLanes lanes(n_lanes);
                        for (int round = 0; round < 3; ++round) {
                            lanes.processAllLasersOnDevice(img_segment_, f_extract_, v_laser_cloud_in, N_SCANS, true, dev, pose_ext);
                            const std::pair<int, int> got = downsampleFusedScans(dev, 0.4f, 0.2f, pose_ext, true);
                            ++tried;
                            agree += got == want ? 1 : 0;
                        }
                    }
                    lv.push_back(agree == tried && want.first > 100 && want.second > 20 ? 1 : 0);
                    std::printf("front end without a host hop: one Device %d + %d thinned features; lanes agree %d / %d\n", want.first, want.second, agree, tried);
                }
                write_file(d + "out_lanes.i32", lv);
                std::printf("front-end lanes from one thread: LiDARs equal on 2 lanes %d / 4, on 4 lanes %d / 4; a job's exception rethrown at wait: %d; sendAhead: same clouds %d, uploads served by it %d\n", lv[0], lv[1], lv[2], lv[3], lv[4]);
            }
            std::printf("re-entrant front end: %d LiDARs on %d threads, one FeatureExtract + one ImageSegmenter: clouds equal %d %d %d %d, labels equal %d %d %d %d\n", NUM_OF_LASER,
                        distinct_threads, verdict[0], verdict[3], verdict[6], verdict[9], verdict[1], verdict[4], verdict[7], verdict[10]);
        }
        // --- round 4: PipelinedMapper -- the overlap's precondition as code. Eight frames of a sensor moving 0.34 m per frame along x (the features of frame k are
        //     the harness's features moved by the inverse motion), odometry that drifts 0.03 m per frame, keyframes every metre: the pipelined loop stages frame k's
        //     maps beside frame k - 1's solve when k - 1 is predicted not to become a keyframe, waits when it is, and once predicts wrongly (frame 3: prior 0.99 m,
        //     result 1.02 m from the last keyframe) -- that frame is solved again on the rebuilt map. Every pose must equal the plain synchronous loop's.
        {
            const int n_frames = 8;
            const Pose T0 = pose;                                          // the converged pose of the scan2map leg above: where the harness's features were taken
            auto motion = [](double dx) { Pose m; m.t_(0) = dx; return m; };
            auto moved = [](const PointICovCloud &c, double dx) { PointICovCloud o = c; for (auto &q : o.points) q.x = float(double(q.x) - dx); return o; };   // M^-1 p, M = translation
            std::vector<PointICovCloud> fs, fc;
            std::vector<Pose> wodom;
            for (int k = 0; k < n_frames; ++k) { fs.push_back(moved(surf, 0.34 * k)); fc.push_back(moved(corner, 0.34 * k)); wodom.push_back(motion(0.31 * k)); }
            // local-map assembly for a keyframe selection: the base maps thinned by a rule that depends on the selection, so that a stale map shows in the pose
            auto assemble = [&](const std::vector<int> &ids, const Pose &, PointICovCloud &s_out, PointICovCloud &c_out) {
                int key = 0;
                for (int id : ids) key += id + 1;
                s_out.clear(); c_out.clear();
                for (size_t i = 0; i < surf_map.size(); ++i) if (int(i % 13) != key % 13) s_out.push_back(surf_map.points[i]);
                for (size_t i = 0; i < corner_map.size(); ++i) if (int(i % 13) != key % 13) c_out.push_back(corner_map.points[i]);
            };
            // (a) the reference's order, one frame at a time (cpp:1065-1101): prior, rebuild after a keyframe, index, solve, transformUpdate, saveKeyframe
            std::vector<Pose> ref_pose;
            std::vector<int> ref_kf;
            {
                KeyframePolicy kf(1.0f, 10.0f, 50.0f);
                FramePipeline one(dev, 3);
                PointICovCloud ms = surf_map, mc = corner_map;
                Pose wmap_wodom = poseMul(T0, poseInverse(wodom[0]));
                bool rebuild = false;
                for (int k = 0; k < n_frames; ++k) {
                    const Pose prior = poseMul(wmap_wodom, wodom[k]);
                    if (rebuild) { PointICovCloud a, b; assemble(kf.surrounding(prior), prior, a, b); ms = a; mc = b; }
                    one.setInputClouds(ms, mc);
                    one.setFeatures(fs[k], fc[k]);
                    one.submit(prior);
                    const Pose r = one.collect();
                    ref_pose.push_back(r);
                    wmap_wodom = poseMul(r, poseInverse(wodom[k]));
                    rebuild = kf.save(r) >= 0;
                    if (rebuild) ref_kf.push_back(k);
                }
            }
            // (b) pipelined
            std::vector<Pose> got_pose;
            std::vector<int> got_kf_index;
            KeyframePolicy kf(1.0f, 10.0f, 50.0f);
            PipelinedMapper mapper(dev, kf, assemble, [&](int idx, const Pose &) { got_kf_index.push_back(idx); }, 3);
            mapper.setInitialMap(surf_map, corner_map);
            mapper.setInitialPose(T0, wodom[0]);
            for (int k = 0; k < n_frames; ++k) {
                Pose prev;
                if (mapper.process(fs[k], fc[k], wodom[k], prev)) got_pose.push_back(prev);
            }
            got_pose.push_back(mapper.finish());
            double worst = 0.0;
            std::vector<double> pm;
            for (int k = 0; k < n_frames; ++k) {
                double a[7], b[7];
                ref_pose[k].toParam(a); got_pose[k].toParam(b);
                for (int i = 0; i < 7; ++i) { worst = std::max(worst, std::fabs(a[i] - b[i])); pm.push_back(b[i]); }
            }
            pm.push_back(worst);
            pm.push_back(double(mapper.counters.overlapped)); pm.push_back(double(mapper.counters.waited)); pm.push_back(double(mapper.counters.redone));
            pm.push_back(double(mapper.counters.keyframes)); pm.push_back(double(ref_kf.size()));
            pm.push_back(double(got_pose[3].t_(0) - got_pose[0].t_(0)));
            write_file(d + "out_pipelined_mapper.f64", pm);
            std::printf("pipelined mapper: %d frames, staged beside the solve %d, waited for the pose %d, solved again after a wrong prediction %d; keyframes %d (plain loop %zu); "
                        "max |pose - plain loop| %.2e\n", n_frames, mapper.counters.overlapped, mapper.counters.waited, mapper.counters.redone, mapper.counters.keyframes,
                        ref_kf.size(), worst);
        }
        // --- the same comparison on RANDOM frame sequences (six seeds x 40 frames): random steps of 0.05-0.6 m, random odometry drift, keyframe distances 0.5-1.5 m --
        //     every mix of "staged beside the solve", "waited for the pose" and "solved again after a wrong prediction" the loop can meet; every pose the plain loop's
        {
            double worst_all = 0.0;
            int tot_overlapped = 0, tot_waited = 0, tot_redone = 0, tot_kf = 0, kf_mismatch = 0;
            for (unsigned seed = 1; seed <= 6; ++seed) {
                std::mt19937 rng(seed);
                std::uniform_real_distribution<double> step(0.05, 0.6), drift(-0.04, 0.04), kfd(0.5, 1.5);
                const int n_frames = 40;
                const float kf_dist = float(kfd(rng));
                const Pose T0 = pose;
                auto motion = [](double dx) { Pose m; m.t_(0) = dx; return m; };
                auto moved = [](const PointICovCloud &c, double dx) { PointICovCloud o = c; for (auto &q : o.points) q.x = float(double(q.x) - dx); return o; };
                std::vector<PointICovCloud> fs, fc;
                std::vector<Pose> wodom;
                double x_true = 0.0, x_odom = 0.0;
                for (int k = 0; k < n_frames; ++k) {
                    fs.push_back(moved(surf, x_true)); fc.push_back(moved(corner, x_true)); wodom.push_back(motion(x_odom));
                    const double st = step(rng);
                    x_true += st; x_odom += st + drift(rng);
                    if (x_true > 6.0) { x_true = 0.0; x_odom = x_odom - 6.0; }                     // stay inside the map
                }
                auto assemble = [&](const std::vector<int> &ids, const Pose &, PointICovCloud &s_out, PointICovCloud &c_out) {
                    int key = 0;
                    for (int id : ids) key += id + 1;
                    s_out.clear(); c_out.clear();
                    for (size_t i = 0; i < surf_map.size(); ++i) if (int(i % 13) != key % 13) s_out.push_back(surf_map.points[i]);
                    for (size_t i = 0; i < corner_map.size(); ++i) if (int(i % 13) != key % 13) c_out.push_back(corner_map.points[i]);
                };
                std::vector<Pose> ref_pose;
                int ref_kf = 0;
                {
                    KeyframePolicy kf(kf_dist, 10.0f, 50.0f);
                    FramePipeline one(dev, 3);
                    PointICovCloud ms = surf_map, mc = corner_map;
                    Pose wmap_wodom = poseMul(T0, poseInverse(wodom[0]));
                    bool rebuild = false;
                    for (int k = 0; k < n_frames; ++k) {
                        const Pose prior = poseMul(wmap_wodom, wodom[k]);
                        if (rebuild) { PointICovCloud a, b; assemble(kf.surrounding(prior), prior, a, b); ms = a; mc = b; }
                        one.setInputClouds(ms, mc);
                        one.setFeatures(fs[k], fc[k]);
                        one.submit(prior);
                        const Pose r = one.collect();
                        ref_pose.push_back(r);
                        wmap_wodom = poseMul(r, poseInverse(wodom[k]));
                        rebuild = kf.save(r) >= 0;
                        ref_kf += rebuild ? 1 : 0;
                    }
                }
                std::vector<Pose> got_pose;
                KeyframePolicy kf(kf_dist, 10.0f, 50.0f);
                PipelinedMapper mapper(dev, kf, assemble, [&](int, const Pose &) {}, 3);
                mapper.setInitialMap(surf_map, corner_map);
                mapper.setInitialPose(T0, wodom[0]);
                for (int k = 0; k < n_frames; ++k) {
                    Pose prev;
                    if (mapper.process(fs[k], fc[k], wodom[k], prev)) got_pose.push_back(prev);
                }
                got_pose.push_back(mapper.finish());
                for (int k = 0; k < n_frames; ++k) {
                    double a[7], b[7];
                    ref_pose[k].toParam(a); got_pose[k].toParam(b);
                    for (int i = 0; i < 7; ++i) worst_all = std::max(worst_all, std::fabs(a[i] - b[i]));
                }
                tot_overlapped += mapper.counters.overlapped; tot_waited += mapper.counters.waited; tot_redone += mapper.counters.redone; tot_kf += mapper.counters.keyframes;
                kf_mismatch += (mapper.counters.keyframes != ref_kf) ? 1 : 0;
            }
            std::vector<double> pr = {worst_all, double(tot_overlapped), double(tot_waited), double(tot_redone), double(tot_kf), double(kf_mismatch)};
            write_file(d + "out_pipelined_mapper_random.f64", pr);
            std::printf("pipelined mapper, 6 random sequences x 40 frames: staged beside the solve %d, waited %d, solved again %d, keyframes %d (sequences whose keyframe count differs "
                        "from the plain loop's: %d); max |pose - plain loop| %.2e\n", tot_overlapped, tot_waited, tot_redone, tot_kf, kf_mismatch, worst_all);
        }
        // --- PoseLocalParameterization sanity
        PoseLocalParameterization lp;
        lp.setParameter();
        double dx[6] = {0.01, 0, 0, 0, 0, 0.02}, xp[7];
        lp.Plus(out, dx, xp);
        std::printf("plus ok %d\n", (int)(xp[0] > out[0]));
    } catch (const std::exception &e) {
        std::fprintf(stderr, "facade_selftest failed: %s\n", e.what());
        return 1;
    }
    return 0;
}
