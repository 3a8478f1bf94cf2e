// blocking.h -- LD blocks from forward correlation row sums (`mps block`).
//
// Own implementation of what /root/reference/cusk/src/blocking.cpp computes: the row-sum profile is
// smoothed with a Hanning window (weights 0.5 - 0.5 cosf(2 pi i / (w - 1)), single-precision cosine, double
// accumulation in window order -- on the device, see cusk_hanning_smooth), blocks are cut at the strict local minima of the smoothed curve that follow
// a higher value (blocking.cpp:37-56), and the odd window size is bisected between 3 and the number of
// markers until the largest block is within 100 of (and not above) the requested maximum (blocking.cpp:85-136).
#pragma once
#include <cmath>
#include <cstdlib>
#include <string>
#include <vector>

namespace host {

struct ChrBlock
{
    size_t first = 0, last = 0;  // chromosome-local marker indices, inclusive
    size_t size() const { return last - first + 1; }
};

// window weights; the O(n * window) smoothing itself runs on the device (cusk_hanning_smooth) through `smooth`
inline std::vector<double> hanning_weights(int window)
{
    std::vector<double> weight((size_t)std::max(window, 0));
    for (int i = 0; i < window; i++) weight[i] = 0.5 - 0.5 * cosf(2.0 * M_PI * (double)i / ((double)window - 1.0));
    return weight;
}

inline std::vector<ChrBlock> cut_at_minima(const std::vector<double> &s)
{
    std::vector<ChrBlock> blocks;
    const long long n = (long long)s.size();
    size_t start = 0;
    double running_max = 0.0;  // highest value since the last cut
    for (long long i = 1; i < n - 1; i++)
    {
        if (running_max > s[(size_t)i] && s[(size_t)i] < s[(size_t)i + 1])
        {
            blocks.push_back({start, (size_t)i});
            start = (size_t)i + 1;
            running_max = 0.0;
        }
        else if (s[(size_t)i] > running_max)
            running_max = s[(size_t)i];
    }
    blocks.push_back({start, (size_t)(n - 1)});
    return blocks;
}

inline int odd_below(int v) { return (v % 2 == 0) ? v - 1 : v; }

// smooth(row_sums, weights) -> smoothed curve of the same length
template <typename Smooth>
inline std::vector<ChrBlock> blocks_of_chromosome(const std::vector<float> &row_sums, int max_block_size, Smooth smooth)
{
    const int tolerance = 100;
    int hi = (int)row_sums.size(), lo = 3;
    int window = odd_below((hi + lo) / 2);
    auto run = [&](int w, int &largest) {
        std::vector<ChrBlock> b = cut_at_minima(smooth(row_sums, hanning_weights(w)));
        size_t big = 0;
        for (const ChrBlock &x : b) big = std::max(big, x.size());
        largest = (int)big;
        return b;
    };
    int largest = 0;
    std::vector<ChrBlock> res = run(window, largest);
    while (std::abs(largest - max_block_size) > tolerance || largest > max_block_size)
    {
        if (largest > max_block_size)
            hi = std::min(hi, window);
        else
            lo = std::max(lo, window);
        const int next = odd_below((hi + lo) / 2);
        if (next == window) break;
        window = next;
        res = run(window, largest);
    }
    return res;
}

}  // namespace host
