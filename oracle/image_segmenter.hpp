// TEST INFRASTRUCTURE ONLY. CPU restatement of ImageSegmenter (estimator/src/imageSegmenter/image_segmenter.hpp:88-393, image_segmenter.cpp:18-61),
// the producer of the ring-major cloud + ScanInfo that FeatureExtract::extractCloud consumes. Pinned against the reference's
// own lines (oracle/_ref, tests/test_oracle_ref_pin.py).
//
// The reference has three spots of undefined behaviour; what this restatement (and the HIP path) does there, stated once:
//   (U1) image_segmenter.hpp:285-286 computes `dist` from `alpha` BEFORE assigning alpha for this neighbour, i.e. with the value the previous
//        neighbour left behind -- and on the very first neighbour of a segmentCloud call with an uninitialised stack slot. `alpha` is declared
//        inside the seed loop; here it is ONE variable that persists over the whole call (what a stack slot does) and starts at 0.
//   (U2) hpp:374 erases outliers with the positions recorded while the scan rows were filled; an earlier erasure in the same row shifts the
//        later points, so a later erasure removes a different point, and may point past the end (UB in std::vector::erase). Here: the stale
//        position is used as it is; one that is no longer inside the row erases nothing.
//   (U3) 64-ring setup: the ground loop starts at ground_scan_id_ = 63 and reads row 64 (hpp:183-185), and segment_alphay_ is never set
//        (image_segmenter.cpp:51-60). The ground loop is clipped to rows that exist; alphay takes the value the BFS assigns (hpp:273-279).
// (Not UB, but easy to misread: hpp:297-299 reads queue_indy_last_negi / queue_last_dis at queue_start_ind AFTER its increment, i.e. the record
//  of the NEXT queue entry -- or, when the queue is otherwise empty, whatever an earlier cluster left at that index; the arrays live for the
//  whole call and start zeroed. Restated literally.)
#pragma once
#include <vector>

namespace orc {

struct SegParams {
    int vertical_scans = 16, horizon_scans = 1800, min_cluster_size = 30, segment_valid_point_num = 5, segment_valid_line_num = 3;
    float segment_theta = 1.047f;    // SEGMENT_THETA (parameters.cpp:39 float)
    double roi_range = 1.0;          // ROI_RANGE (parameters.cpp:58 double)
    bool segment_flag = true;        // ScanInfo::segment_flag_
};

struct SegResult {
    std::vector<float> cloud_out;        // n x 4 ring-major [x y z intensity + row]
    std::vector<float> cloud_outlier;    // m x 4
    std::vector<int> scan_start, scan_end;
    std::vector<float> range_mat;        // vs x hs (FLT_MAX = empty)
    std::vector<int> label_mat;          // vs x hs after the BFS (-1 empty, 1 ground, >= 2 clusters, 999999 outlier)
    std::vector<int> pixel_of_point;     // per input point: row * hs + col of the pixel it won, or -1
};

void segment_cloud(const float *xyzi, int n, const SegParams &prm, SegResult &out);

}  // namespace orc
