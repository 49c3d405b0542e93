// "A std::vector that only ever loses elements", answered in O(log n): element at position j, membership, erase -- shared by the feature selection
// loops (select.hip: all_feature_idx, lidar_mapper.h:350, 531-553) and the segmenter's outlier erasure (segment.hip: cloud_scan[row].erase,
// image_segmenter.hpp:366-376). Host code.
#pragma once
#include <cstddef>
#include <cstdint>
#include <vector>

namespace mlh {

// The reference keeps the not-yet-consumed feature slots in a std::vector it erases from (all_feature_idx, lidar_mapper.h:350,
// 531-553): position j of that vector is always the (j+1)-th surviving ORIGINAL index, because it starts as 0..M-1 and only ever
// loses elements. The same three questions -- element at position j, position of an element (the reference's std::find), erase --
// are answered here from one "alive" bit per slot under a 16-ary tree of populations, with identical results. Each tree node's sixteen
// children keep their EXCLUSIVE prefix sums side by side (one 64-byte line): going down a level is "how many of the fifteen prefixes are
// <= k" -- sixteen independent compares the compiler turns into four SSE2 ones, no branch (the outcome is a coin flip) -- and an erase is
// "subtract one from the siblings to the right", a masked vector subtract per level. The draw loops are one dependent chain per draw --
// draw, look up, erase -- so the look-up's latency is the loop's speed: a Fenwick tree over single slots with a data-dependent branch per
// level ran at ~170 ns per draw, a branch-free Fenwick tree over the words (eight dependent loads for 12 k slots) at ~30, this (two levels
// for 12 k slots, three up to 262 k) at roughly half of that. The rank search inside the word is unchanged.
class AlivePool {
public:
    struct Select8 {                                               // at[v][r] = position of the r-th (0-based) set bit of the byte v
        uint8_t at[256][8];
        Select8()
        {
            for (int v = 0; v < 256; ++v) {
                int r = 0;
                for (int b = 0; b < 8; ++b) if (v >> b & 1) at[v][r++] = uint8_t(b);
                for (; r < 8; ++r) at[v][r] = 0;
            }
        }
    };
    static inline const Select8 kSelect8{};
    struct After {                                                 // m[c][s] = 1 for the siblings to the right of child c
        int32_t m[16][16];
        After() { for (int c = 0; c < 16; ++c) for (int s = 0; s < 16; ++s) m[c][s] = s > c; }
    };
    static inline const After kAfter{};
    explicit AlivePool(size_t n) : alive_(n), nw_((n + 63) / 64)
    {
        bits_.assign(nw_ + 1, 0);
        for (size_t w = 0; w < nw_; ++w) bits_[w] = (w * 64 + 64 <= n) ? ~uint64_t(0) : ((uint64_t(1) << (n - w * 64)) - 1);
        // level 0: one node per word; every level above: one node per group of sixteen below; siblings that do not exist hold kNever
        std::vector<int32_t> cnt(nw_ ? nw_ : 1, 0);
        for (size_t w = 0; w < nw_; ++w) cnt[w] = int32_t(popcount64(bits_[w]));
        size_t nodes = cnt.size();
        levels_ = 0;
        do {
            const size_t groups = (nodes + 15) / 16;
            off_[levels_] = pre_.size();
            pre_.resize(pre_.size() + groups * 16, kNever);
            std::vector<int32_t> up(groups);
            for (size_t g = 0; g < groups; ++g) {
                int32_t run = 0;
                for (size_t s = 0; s < 16 && g * 16 + s < nodes; ++s) { pre_[off_[levels_] + g * 16 + s] = run; run += cnt[g * 16 + s]; }
                up[g] = run;
            }
            cnt.swap(up);
            nodes = groups;
            ++levels_;
        } while (nodes > 1);
    }
    size_t size() const { return alive_; }
    bool empty() const { return alive_ == 0; }
    // original index of the element at position j (0-based) among the survivors
    __attribute__((always_inline)) size_t at(size_t j) const
    {
        int32_t k = int32_t(j);
        size_t node = 0;
        for (int l = levels_ - 1; l >= 0; --l) {
            const int32_t *p = &pre_[off_[l] + node * 16];
            int c = 0;
            for (int s = 1; s < 16; ++s) c += (p[s] <= k);           // prefixes are non-decreasing: the last child whose prefix is <= k (empty children are stepped over)
            k -= p[c];
            node = node * 16 + size_t(c);
        }
        const size_t pos = node;
        // rank search inside the word: per-byte populations (SWAR), their running sums by one multiply, the first byte whose running sum
        // exceeds k by a carry-free byte-wise compare, the bit inside that byte from a 2 KB table
        const uint64_t w = bits_[pos];
        uint64_t c = w - ((w >> 1) & 0x5555555555555555ull);
        c = (c & 0x3333333333333333ull) + ((c >> 2) & 0x3333333333333333ull);
        c = (c + (c >> 4)) & 0x0f0f0f0f0f0f0f0full;
        const uint64_t run = c * 0x0101010101010101ull;               // byte i: population of bytes 0..i (<= 64)
        const uint64_t over = ((run | 0x8080808080808080ull) - (uint64_t(k) + 1) * 0x0101010101010101ull) & 0x8080808080808080ull;
        const unsigned byte = unsigned(__builtin_ctzll(over)) >> 3;   // first byte with run > k (exists: k < the word's population)
        const unsigned before = unsigned(((run << 8) >> (8 * byte)) & 0xff);
        const size_t bit = 8 * byte + kSelect8.at[(w >> (8 * byte)) & 0xff][unsigned(k) - before];
        return pos * 64 + bit;
    }
    bool contains(size_t idx) const { return (bits_[idx >> 6] >> (idx & 63)) & 1; }
    __attribute__((always_inline)) void erase_index(size_t idx)
    {
        bits_[idx >> 6] &= ~(uint64_t(1) << (idx & 63));
        --alive_;
        size_t node = idx >> 6;
        for (int l = 0; l < levels_; ++l) {
            int32_t *p = &pre_[off_[l] + (node & ~size_t(15))];
            const int32_t *m = kAfter.m[node & 15];
            for (int s = 0; s < 16; ++s) p[s] -= m[s];
            node >>= 4;
        }
    }
private:
    static constexpr int32_t kNever = 0x3fffffff;                  // never <= k, and still not after every slot has been erased
    static inline unsigned popcount64(uint64_t x)                  // SWAR: the baseline x86-64 target has no popcnt instruction
    {
        x = x - ((x >> 1) & 0x5555555555555555ull);
        x = (x & 0x3333333333333333ull) + ((x >> 2) & 0x3333333333333333ull);
        x = (x + (x >> 4)) & 0x0f0f0f0f0f0f0f0full;
        return unsigned((x * 0x0101010101010101ull) >> 56);
    }
    size_t alive_, nw_;
    std::vector<uint64_t> bits_;
    std::vector<int32_t> pre_;      // all levels, one after the other (off_[l]), sixteen exclusive prefix sums per node group
    size_t off_[16] = {0};
    int levels_ = 0;
};


}  // namespace mlh
