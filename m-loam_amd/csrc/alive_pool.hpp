// "A std::vector that only ever loses elements", answered in O(log n): element at position j, membership, erase -- shared by the feature selection
// loops (select.hip: all_feature_idx, lidar_mapper.h:350, 531-553) and the segmenter's outlier erasure (segment.hip: cloud_scan[row].erase,
// image_segmenter.hpp:366-376). Host code.
#pragma once
#include <cstddef>
#include <cstdint>
#include <vector>
#if defined(__SSE2__)
#include <emmintrin.h>
#endif

namespace mlh {

// The reference keeps the not-yet-consumed feature slots in a std::vector it erases from (all_feature_idx, lidar_mapper.h:350,
// 531-553): position j of that vector is always the (j+1)-th surviving ORIGINAL index, because it starts as 0..M-1 and only ever
// loses elements. The same three questions -- element at position j, position of an element (the reference's std::find), erase --
// are answered here, with identical results, from a tree of EXCLUSIVE PREFIX COUNTS kept side by side per node: a leaf is 64 slots with
// one byte each (how many survivors to the left inside the leaf: one 64-byte line), above it groups of sixteen leaves with int16
// prefixes, above those groups of sixteen with int32 prefixes. Prefixes never decrease from left to right, so "which child holds the
// k-th survivor" is "where does the prefix first exceed k": a vector compare, a move-mask and a count-trailing-zeros per level (SSE2, which
// every x86-64 has; plain loops otherwise), no branch -- the outcome is a coin flip -- and an erase is "subtract one from everything to
// the right", a compare against the lane numbers added to the line. The draw loops are one dependent chain per draw -- draw, look up,
// erase -- so the look-up's cost is the loop's speed: a Fenwick tree over single slots with a data-dependent branch per level ran at
// ~170 ns per draw, a branch-free Fenwick tree over 64-bit words at ~30, a 16-ary tree over the words with a SWAR rank search inside the
// word at ~20; this form needs three dependent loads for 12 k slots and no arithmetic rank search (scripts/exp/select_loop_bench.sh).
class AlivePool {
public:
    explicit AlivePool(size_t n) : alive_(n), nl_((n + 63) / 64 ? (n + 63) / 64 : 1)
    {
        bits_.assign(nl_, 0);
        leaf_.resize(nl_);
        std::vector<int32_t> cnt(nl_, 0);
        for (size_t l = 0; l < nl_; ++l) {
            int run = 0;
            for (size_t s = 0; s < 64; ++s) {
                leaf_[l].pre[s] = uint8_t(run);
                if (l * 64 + s < n) { bits_[l] |= uint64_t(1) << s; ++run; }
            }
            cnt[l] = run;
        }
        // level 1: sixteen leaves per group, int16 prefixes
        const size_t g1 = (nl_ + 15) / 16;
        l1_.resize(g1);
        std::vector<int32_t> up(g1);
        for (size_t g = 0; g < g1; ++g) {
            int32_t run = 0;
            for (size_t s = 0; s < 16; ++s) {
                if (g * 16 + s < nl_) { l1_[g].pre[s] = int16_t(run); run += cnt[g * 16 + s]; } else l1_[g].pre[s] = 0x7fff;
            }
            up[g] = run;
        }
        cnt.swap(up);
        // levels above: sixteen children per group, int32 prefixes
        size_t nodes = g1;
        levels_ = 0;
        while (nodes > 1) {
            const size_t groups = (nodes + 15) / 16;
            off_[levels_] = pre_.size();
            pre_.resize(pre_.size() + groups * 16, kNever);
            std::vector<int32_t> u2(groups);
            for (size_t g = 0; g < groups; ++g) {
                int32_t run = 0;
                for (size_t s = 0; s < 16 && g * 16 + s < nodes; ++s) { pre_[off_[levels_] + g * 16 + s] = run; run += cnt[g * 16 + s]; }
                u2[g] = run;
            }
            cnt.swap(u2); nodes = groups; ++levels_;
        }
    }
    size_t size() const { return alive_; }
    bool empty() const { return alive_ == 0; }
    __attribute__((always_inline)) size_t at(size_t j) const
    {
        int32_t k = int32_t(j);
        size_t node = 0;
        for (int l = levels_ - 1; l >= 0; --l) {
            const int32_t *p = &pre_[off_[l] + node * 16];
            int c = 0;
            for (int s = 1; s < 16; ++s) c += (p[s] <= k);
            k -= p[c];
            node = node * 16 + size_t(c);
        }
        {   // level 1
            const int16_t *p = l1_[node].pre;
#if defined(__SSE2__)
            const __m128i kk = _mm_set1_epi16(short(k));
            const __m128i g0 = _mm_cmpgt_epi16(_mm_load_si128(reinterpret_cast<const __m128i *>(p)), kk);
            const __m128i g1 = _mm_cmpgt_epi16(_mm_load_si128(reinterpret_cast<const __m128i *>(p + 8)), kk);
            const unsigned m = unsigned(_mm_movemask_epi8(_mm_packs_epi16(g0, g1)));
            const int c = (m ? __builtin_ctz(m) : 16) - 1;
#else
            int c = -1; for (int s = 0; s < 16; ++s) c += (p[s] <= k);
#endif
            k -= p[c];
            node = node * 16 + size_t(c);
        }
        const uint8_t *p = leaf_[node].pre;
#if defined(__SSE2__)
        const __m128i kk = _mm_set1_epi8(char(k));
        const uint64_t m0 = unsigned(_mm_movemask_epi8(_mm_cmpgt_epi8(_mm_load_si128(reinterpret_cast<const __m128i *>(p)), kk)));
        const uint64_t m1 = unsigned(_mm_movemask_epi8(_mm_cmpgt_epi8(_mm_load_si128(reinterpret_cast<const __m128i *>(p + 16)), kk)));
        const uint64_t m2 = unsigned(_mm_movemask_epi8(_mm_cmpgt_epi8(_mm_load_si128(reinterpret_cast<const __m128i *>(p + 32)), kk)));
        const uint64_t m3 = unsigned(_mm_movemask_epi8(_mm_cmpgt_epi8(_mm_load_si128(reinterpret_cast<const __m128i *>(p + 48)), kk)));
        const uint64_t m = m0 | (m1 << 16) | (m2 << 32) | (m3 << 48);
        const int c = (m ? __builtin_ctzll(m) : 64) - 1;
#else
        int c = -1; for (int s = 0; s < 64; ++s) c += (p[s] <= k);
#endif
        return node * 64 + size_t(c);
    }
    bool contains(size_t idx) const { return (bits_[idx >> 6] >> (idx & 63)) & 1; }
    __attribute__((always_inline)) void erase_index(size_t idx)
    {
        const size_t lf = idx >> 6;
        bits_[lf] &= ~(uint64_t(1) << (idx & 63));
        --alive_;
        {
            uint8_t *p = leaf_[lf].pre;
#if defined(__SSE2__)
            const __m128i sl = _mm_set1_epi8(char(idx & 63));
            const __m128i i0 = _mm_setr_epi8(0,1,2,3,4,5,6,7,8,9,10,11,12,13,14,15), st = _mm_set1_epi8(16);
            __m128i it = i0;
            for (int r = 0; r < 4; ++r) {
                __m128i *q = reinterpret_cast<__m128i *>(p + 16 * r);
                _mm_store_si128(q, _mm_add_epi8(_mm_load_si128(q), _mm_cmpgt_epi8(it, sl)));
                it = _mm_add_epi8(it, st);
            }
#else
            for (size_t s = (idx & 63) + 1; s < 64; ++s) --p[s];
#endif
        }
        {
            int16_t *p = l1_[lf >> 4].pre;
#if defined(__SSE2__)
            const __m128i ch = _mm_set1_epi16(short(lf & 15));
            __m128i *q0 = reinterpret_cast<__m128i *>(p), *q1 = reinterpret_cast<__m128i *>(p + 8);
            _mm_store_si128(q0, _mm_add_epi16(_mm_load_si128(q0), _mm_cmpgt_epi16(_mm_setr_epi16(0,1,2,3,4,5,6,7), ch)));
            _mm_store_si128(q1, _mm_add_epi16(_mm_load_si128(q1), _mm_cmpgt_epi16(_mm_setr_epi16(8,9,10,11,12,13,14,15), ch)));
#else
            for (size_t s = (lf & 15) + 1; s < 16; ++s) --p[s];
#endif
        }
        size_t node = lf >> 4;
        for (int l = 0; l < levels_; ++l) {
            int32_t *p = &pre_[off_[l] + (node & ~size_t(15))];
            const int c = int(node & 15);
            for (int s = 0; s < 16; ++s) p[s] -= (s > c);
            node >>= 4;
        }
    }
private:
    static constexpr int32_t kNever = 0x3fffffff;
    struct alignas(64) Leaf { uint8_t pre[64]; };
    struct alignas(32) L1 { int16_t pre[16]; };
    size_t alive_, nl_;
    std::vector<uint64_t> bits_;
    std::vector<Leaf> leaf_;
    std::vector<L1> l1_;
    std::vector<int32_t> pre_;
    size_t off_[16] = {0};
    int levels_ = 0;
};


}  // namespace mlh
