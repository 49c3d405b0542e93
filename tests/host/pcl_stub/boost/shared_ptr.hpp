// TEST INFRASTRUCTURE: boost::shared_ptr / make_shared over the standard ones (pcl 1.8's Ptr types are boost::shared_ptr)
#pragma once
#include <memory>
#include <utility>
namespace boost {
using std::shared_ptr;
template <typename T, typename... A> std::shared_ptr<T> make_shared(A &&...a) { return std::make_shared<T>(std::forward<A>(a)...); }
}  // namespace boost
