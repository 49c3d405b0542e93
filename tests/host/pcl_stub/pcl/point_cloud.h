// TEST INFRASTRUCTURE: pcl::PointCloud<PointT> as the hot path's call sites use it (pcl/point_cloud.h): public `points`, width / height / is_dense, size,
// push_back (keeps width = size, height = 1, as PCL does), clear, operator[], begin / end, operator+=, Ptr / ConstPtr (boost::shared_ptr), makeShared.
#pragma once
#include <cstddef>
#include <cstdint>
#include <vector>
#include <boost/shared_ptr.hpp>
namespace pcl {
struct PCLHeader { std::uint32_t seq = 0; std::uint64_t stamp = 0; };
template <typename PointT>
class PointCloud {
public:
    typedef PointT PointType;
    typedef boost::shared_ptr<PointCloud<PointT>> Ptr;
    typedef boost::shared_ptr<const PointCloud<PointT>> ConstPtr;
    typedef typename std::vector<PointT>::iterator iterator;
    typedef typename std::vector<PointT>::const_iterator const_iterator;
    PCLHeader header;
    std::vector<PointT> points;
    std::uint32_t width = 0, height = 0;
    bool is_dense = true;
    PointCloud() {}
    size_t size() const { return points.size(); }
    bool empty() const { return points.empty(); }
    void resize(size_t n) { points.resize(n); if (width * height != n) { width = std::uint32_t(n); height = 1; } }
    void reserve(size_t n) { points.reserve(n); }
    void push_back(const PointT &p) { points.push_back(p); width = std::uint32_t(points.size()); height = 1; }
    void clear() { points.clear(); width = 0; height = 0; }
    PointT &operator[](size_t i) { return points[i]; }
    const PointT &operator[](size_t i) const { return points[i]; }
    iterator begin() { return points.begin(); }
    iterator end() { return points.end(); }
    const_iterator begin() const { return points.begin(); }
    const_iterator end() const { return points.end(); }
    PointCloud &operator+=(const PointCloud &o) { points.insert(points.end(), o.points.begin(), o.points.end()); width = std::uint32_t(points.size()); height = 1; return *this; }
    Ptr makeShared() const { return Ptr(new PointCloud<PointT>(*this)); }
};
}  // namespace pcl
