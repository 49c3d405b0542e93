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
// are answered here from one "alive" bit per slot plus a Fenwick tree over the 64-bit words' populations: a branch-free descent over
// log2(M/64) levels and a six-step rank search inside the word, instead of O(M) per operation, with identical results. (The draw loops
// are one dependent chain per draw -- draw, look up, erase -- so the lookup's latency is the loop's speed: a Fenwick tree over single
// slots with a data-dependent branch per level ran at ~170 ns per draw, this runs at ~30.)
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
    explicit AlivePool(size_t n) : alive_(n), nw_((n + 63) / 64)
    {
        log_ = 0;
        while ((size_t(2) << log_) <= nw_) ++log_;                  // largest power of two <= nw_ is 1 << log_
        bits_.assign(nw_ + 1, 0);
        for (size_t w = 0; w < nw_; ++w) bits_[w] = (w * 64 + 64 <= n) ? ~uint64_t(0) : ((uint64_t(1) << (n - w * 64)) - 1);
        t_.assign((size_t(2) << log_) + 1, kNever);                  // slots past nw_ are never taken by the descent
        for (size_t i = 1; i <= nw_; ++i) t_[i] = 0;
        for (size_t i = 1; i <= nw_; ++i) {
            t_[i] += int32_t(popcount64(bits_[i - 1]));
            const size_t j = i + (i & (~i + 1));
            if (j <= nw_) t_[j] += t_[i];
        }
    }
    size_t size() const { return alive_; }
    bool empty() const { return alive_ == 0; }
    // original index of the element at position j (0-based) among the survivors
    size_t at(size_t j) const
    {
        size_t pos = 0;
        int32_t k = int32_t(j);
        for (int b = log_; b >= 0; --b) {
            const int32_t v = t_[pos + (size_t(1) << b)];
            const int32_t take = -int32_t(v <= k);                  // all-ones / zero: the comparison's outcome is a coin flip, keep it out of the branch predictor
            pos += (size_t(1) << b) & size_t(int64_t(take));
            k -= v & take;
        }
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
    void erase_index(size_t idx)
    {
        bits_[idx >> 6] &= ~(uint64_t(1) << (idx & 63));
        --alive_;
        for (size_t i = (idx >> 6) + 1; i <= nw_; i += i & (~i + 1)) t_[i] -= 1;
    }
private:
    static constexpr int32_t kNever = 0x3fffffff;
    static inline unsigned popcount64(uint64_t x)                  // SWAR: the baseline x86-64 target has no popcnt instruction
    {
        x = x - ((x >> 1) & 0x5555555555555555ull);
        x = (x & 0x3333333333333333ull) + ((x >> 2) & 0x3333333333333333ull);
        x = (x + (x >> 4)) & 0x0f0f0f0f0f0f0f0full;
        return unsigned((x * 0x0101010101010101ull) >> 56);
    }
    size_t alive_, nw_;
    std::vector<uint64_t> bits_;
    std::vector<int32_t> t_;
    int log_;
};


}  // namespace mlh
