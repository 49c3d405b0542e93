// Host logic check (no GPU): the facade's FrontEndLanes (mloam_facade.hpp) -- the workers behind estimator.cpp:248-263 for a caller without OpenMP -- with stand-in
// segmenter / extractor types that record who ran what where. Compiled and run by tests/test_abi.py::test_facade_front_end_lanes.
//   1  processAllLasers: every LiDAR's calTimestamp -> segmentCloud -> extractCloud ran once, in that order, with that LiDAR's cloud, "laser_cloud_outlier" inserted;
//      more LiDARs than lanes go in passes; the lanes really are different threads, none of them the caller's, and the same ones on the next frame
//   2  two jobs posted to two lanes overlap in time (each waits for the other's start)
//   3  a job's exception is rethrown at wait(), the lane takes the next job; processAllLasers rethrows a LiDAR's failure after all lanes are done
//   4  post() behind a running job waits for it (jobs of one lane run in posting order)
#include "mloam_facade.hpp"
#include <atomic>
#include <chrono>
#include <cstdio>
#include <set>

using namespace mloam_hip;

struct Log {
    std::mutex mu;
    std::vector<std::string> events;
    std::set<std::thread::id> threads;
    void add(const std::string &e) { std::lock_guard<std::mutex> g(mu); events.push_back(e); threads.insert(std::this_thread::get_id()); }
};
struct MockExtract {
    Log *log;
    int fail_at = -1;
    void calTimestamp(const PointXYZCloud &in, PointICloud &out) const
    {
        out.points.resize(in.size());
        for (size_t i = 0; i < in.size(); ++i) { out.points[i].x = in.points[i].x; out.points[i].intensity = 0.5f; }
        log->add("cal " + std::to_string(int(in.points.at(0).x)));
    }
    template <class SI> void extractCloud(const PointICloud &in, const SI &si, cloudFeature &cf)
    {
        const int id = int(in.points.at(0).x);
        if (id == fail_at) throw Error("extract failed for " + std::to_string(id));
        cf["laser_cloud"] = in;
        cf["n_scans"].points.resize(si.scan_start_ind_.size());
        log->add("extract " + std::to_string(id));
    }
};
struct MockSegment {
    Log *log;
    template <class SI> void segmentCloud(const PointICloud &in, PointICloud &out, PointICloud &outlier, SI &si)
    {
        out = in;
        outlier.points.resize(3);
        si.scan_start_ind_.assign(size_t(7), si.segment_flag_ ? 1 : 0);
        log->add("segment " + std::to_string(int(in.points.at(0).x)));
        std::this_thread::sleep_for(std::chrono::milliseconds(2));
    }
};

#define CHECK(c) do { if (!(c)) { std::fprintf(stderr, "front_end_lanes_check: line %d: %s\n", __LINE__, #c); return 1; } } while (0)

int main()
{
    // 1
    {
        Log log;
        MockExtract fx{&log};
        MockSegment seg{&log};
        const int L = 5;
        std::vector<PointXYZCloud> in(L);
        for (int l = 0; l < L; ++l) { PointXYZ q; q.x = float(l + 1); in[size_t(l)].push_back(q); in[size_t(l)].push_back(q); }
        FrontEndLanes lanes(2);
        CHECK(lanes.size() == 2);
        std::vector<cloudFeature> ff;
        std::set<std::thread::id> first;
        for (int frame = 0; frame < 3; ++frame) {
            log.events.clear(); log.threads.clear();
            lanes.processAllLasers(seg, fx, in, 16, frame != 1, ff);
            CHECK(int(ff.size()) == L);
            CHECK(int(log.events.size()) == 3 * L);
            for (int l = 1; l <= L; ++l) {                      // per LiDAR: cal < segment < extract
                int pos[3] = {-1, -1, -1};
                for (size_t e = 0; e < log.events.size(); ++e) {
                    if (log.events[e] == "cal " + std::to_string(l)) pos[0] = int(e);
                    if (log.events[e] == "segment " + std::to_string(l)) pos[1] = int(e);
                    if (log.events[e] == "extract " + std::to_string(l)) pos[2] = int(e);
                }
                CHECK(pos[0] >= 0 && pos[0] < pos[1] && pos[1] < pos[2]);
                cloudFeature &cf = ff[size_t(l - 1)];
                CHECK(cf.count("laser_cloud_outlier") == 1 && cf["laser_cloud_outlier"].size() == 3);
                CHECK(cf["laser_cloud"].size() == 2 && int(cf["laser_cloud"].points[0].x) == l);
                CHECK(cf["n_scans"].size() == 7);
            }
            CHECK(log.threads.size() == 2 && log.threads.count(std::this_thread::get_id()) == 0);
            if (frame == 0) first = log.threads;
            else CHECK(log.threads == first);
        }
        // 3b: one LiDAR fails: the others are done, the failure arrives at the caller, the lanes go on
        fx.fail_at = 2;
        bool caught = false;
        log.events.clear();
        try { lanes.processAllLasers(seg, fx, in, 16, true, ff); } catch (const Error &e) { caught = std::string(e.what()).find("extract failed for 2") != std::string::npos; }
        CHECK(caught);
        fx.fail_at = -1;
        lanes.processAllLasers(seg, fx, in, 16, true, ff);
        CHECK(int(ff[1]["laser_cloud"].points[0].x) == 2);
    }
    // 2
    {
        FrontEndLanes lanes(2);
        std::atomic<int> started{0};
        std::atomic<bool> overlapped[2] = {{false}, {false}};
        for (int i = 0; i < 2; ++i)
            lanes.post(i, [&started, &overlapped, i] {
                started.fetch_add(1);
                const auto t0 = std::chrono::steady_clock::now();
                while (started.load() < 2 && std::chrono::steady_clock::now() - t0 < std::chrono::seconds(5)) std::this_thread::yield();
                overlapped[i] = started.load() == 2;
            });
        lanes.wait(0); lanes.wait(1);
        CHECK(overlapped[0] && overlapped[1]);
    }
    // 3a, 4
    {
        FrontEndLanes lanes(1);
        bool caught = false;
        lanes.post(0, [] { throw Error("lane job failed"); });
        try { lanes.wait(0); } catch (const Error &) { caught = true; }
        CHECK(caught);
        lanes.wait(0);                                            // nothing pending: returns, throws nothing
        std::vector<int> order;
        lanes.post(0, [&order] { std::this_thread::sleep_for(std::chrono::milliseconds(5)); order.push_back(1); });
        lanes.post(0, [&order] { order.push_back(2); });          // waits for the first
        lanes.wait(0);
        CHECK(order.size() == 2 && order[0] == 1 && order[1] == 2);
    }
    std::printf("front_end_lanes_check ok\n");
    return 0;
}
