// merge.h -- `merge-block-outputs` on in-memory block results: the merged sparse skeleton files straight from the results
// the multi-GPU job has just gathered to rank 0 (north_star: "an RCCL-over-xGMI gather of the merged adjacency").
//
// Mirrors /root/reference/cusk_postprocessing/merge_blocks.py: merge_block_outputs (:361-395), add_sam / add_scm / add_gmi
// (:328-346), the BlockOutput loaders (:18-72, :99-116) and GlobalBdpcResult.write_mm (:298-325).  The reference keeps the
// merged matrices as Python dicts keyed by (row, column) and writes them in insertion order; an update of an existing key
// keeps its place, a deleted key that comes back goes to the end.  Values are written as an f-string prints a numpy float32:
// repr of the Python float it converts to.
// Same files byte for byte: tests/test_merge_golden.py (the files the reference itself wrote) and
// tests/test_gpu_batch.py (against this package's Python mirror and the oracle's restatement).
#pragma once
#include <charconv>
#include <cstdint>
#include <cstring>
#include <map>
#include <string>
#include <unordered_map>
#include <vector>

#include "block_pipeline.h"

namespace host {

// f"{numpy.float32(v)}" as the reference's write_mm prints it (merge_blocks.py:309-318): numpy scalars format through
// Python's float, i.e. repr(float(v)) -- the shortest digits that round-trip the DOUBLE value, positional when the decimal
// exponent lies in [-4, 16), else scientific with at least two exponent digits
inline std::string python_float_str(float v)
{
    if (v != v) return "nan";
    if (v == std::numeric_limits<float>::infinity()) return "inf";
    if (v == -std::numeric_limits<float>::infinity()) return "-inf";
    if (v == 0.0f) return std::signbit(v) ? "-0.0" : "0.0";
    char buf[64];
    auto r = std::to_chars(buf, buf + sizeof(buf), (double)v, std::chars_format::scientific);  // shortest round-trip digits
    const std::string sci(buf, r.ptr);
    const bool neg = sci[0] == '-';
    const size_t epos = sci.find('e');
    const std::string mant = sci.substr(neg ? 1 : 0, epos - (neg ? 1 : 0));
    const int e10 = std::atoi(sci.c_str() + epos + 1);
    if (e10 < -4 || e10 >= 16) return sci;  // (to_chars writes the exponent as Python does: sign, at least two digits)
    std::string digits;
    for (char c : mant)
        if (c != '.') digits.push_back(c);
    std::string out = neg ? "-" : "";
    if (e10 >= 0)
    {
        const size_t ip = (size_t)e10 + 1;  // digits before the point
        if (digits.size() <= ip)
        {
            out += digits;
            out.append(ip - digits.size(), '0');
            out += ".0";
        }
        else
        {
            out += digits.substr(0, ip);
            out += ".";
            out += digits.substr(ip);
        }
    }
    else
    {
        out += "0.";
        out.append((size_t)(-e10 - 1), '0');
        out += digits;
    }
    return out;
}

// insertion-ordered (row, column) -> value store with the update / delete semantics of a Python dict
template <typename V>
struct OrderedEntries
{
    std::vector<std::pair<uint64_t, V>> items;
    std::vector<char> alive;
    std::unordered_map<uint64_t, size_t> pos;
    static uint64_t key(int64_t i, int64_t j) { return ((uint64_t)i << 32) | (uint32_t)j; }
    void put(int64_t i, int64_t j, V v)
    {
        const uint64_t k = key(i, j);
        auto it = pos.find(k);
        if (it == pos.end())
        {
            pos.emplace(k, items.size());
            items.emplace_back(k, v);
            alive.push_back(1);
        }
        else
            items[it->second].second = v;
    }
    bool has(int64_t i, int64_t j) const { return pos.count(key(i, j)) != 0; }
    void drop(int64_t i, int64_t j)
    {
        auto it = pos.find(key(i, j));
        if (it == pos.end()) return;
        alive[it->second] = 0;
        pos.erase(it);
    }
    size_t size() const { return pos.size(); }
};

struct MergeInput
{
    bool present = false;
    size_t num_var = 0, num_phen = 0, max_level = 0;
    const int *ixs = nullptr, *adj = nullptr;
    const float *corr = nullptr;
};

// blocks: one entry per line of the .blocks file, in file order (absent: the block wrote no files);
// block_sizes: markers of every listed block.  Writes <basepath>_sam.mtx, _scm.mtx, .mdim, .ixs.
inline void merge_blocks_to_files(const std::vector<MergeInput> &blocks, const std::vector<size_t> &block_sizes, const std::string &basepath)
{
    OrderedEntries<int> sam;
    OrderedEntries<float> scm;
    std::map<int64_t, int64_t> gmi;  // merged marker index -> global (.bim row) index
    int64_t sel_off = 0, glob_off = 0;
    size_t last_p = 0, last_ml = 0;
    bool any = false;
    for (size_t index = 0; index < blocks.size(); index++)
    {
        const MergeInput &b = blocks[index];
        if (!b.present)
        {  // merge_blocks.py:376-379, :384-387: the block still shifts the global marker indices
            glob_off += (int64_t)block_sizes[index];
            continue;
        }
        const int64_t nv = (int64_t)b.num_var, np_ = (int64_t)b.num_phen, nm = nv - np_;
        // :24-32: block-local index (markers first, then traits) -> merged 1-based index (traits first)
        auto merged = [&](int64_t d) { return d < nm ? d + sel_off + np_ + 1 : d - nm + 1; };
        if (index == 0)
        {  // :367-373: only the FIRST LISTED block is taken as it is
            for (int64_t r = 0; r < nv; r++)
                for (int64_t c = 0; c < nv; c++)
                    if (b.adj[r * nv + c] != 0) sam.put(merged(r), merged(c), b.adj[r * nv + c]);
        }
        else
        {
            // :336-341 (0-based i, j < num_p against 1-based keys: links of the last trait are never intersected)
            OrderedEntries<int> have;
            for (int64_t r = 0; r < nv; r++)
                for (int64_t c = 0; c < nv; c++)
                    if (b.adj[r * nv + c] != 0) have.put(merged(r), merged(c), 1);
            for (int64_t i = 0; i < np_; i++)
                for (int64_t j = 0; j < np_; j++)
                    if (sam.has(i, j) && !have.has(i, j)) sam.drop(i, j);
            // :343-345
            for (int64_t r = 0; r < nv; r++)
                for (int64_t c = 0; c < nv; c++)
                {
                    const int v = b.adj[r * nv + c];
                    if (v == 0) continue;
                    const int64_t i = merged(r), j = merged(c);
                    if (i >= np_ || j >= np_) sam.put(i, j, v);
                }
        }
        for (int64_t r = 0; r < nv; r++)  // :332-333
            for (int64_t c = 0; c < nv; c++)
            {
                const float v = b.corr[r * nv + c];
                if (v != 0.0f) scm.put(merged(r), merged(c), v);  // (np.nonzero: NaN counts as non-zero)
            }
        for (int64_t d = 0; d < nv; d++)  // :54-72
            if (merged(d) >= np_ + 1) gmi[merged(d)] = (int64_t)b.ixs[d] + glob_off;
        sel_off += nm;
        glob_off += (int64_t)block_sizes[index];
        last_p = b.num_phen;
        last_ml = b.max_level;
        any = true;
    }
    if (!any) die("no block output to merge");
    int64_t dim = 0;  // :300-303: both headers carry the adjacency's largest row index
    for (size_t k = 0; k < sam.items.size(); k++)
        if (sam.alive[k]) dim = std::max<int64_t>(dim, (int64_t)(sam.items[k].first >> 32));
    {
        std::string out = "%%MatrixMarket matrix coordinate integer general\n";
        out += std::to_string(dim) + "\t" + std::to_string(dim) + "\t" + std::to_string(sam.size()) + "\n";
        for (size_t k = 0; k < sam.items.size(); k++)
            if (sam.alive[k])
                out += std::to_string(sam.items[k].first >> 32) + "\t" + std::to_string((uint32_t)sam.items[k].first) + "\t" +
                       std::to_string(sam.items[k].second) + "\n";
        write_binary(basepath + "_sam.mtx", out.data(), out.size());
    }
    {
        std::string out = "%%MatrixMarket matrix coordinate real general\n";
        out += std::to_string(dim) + "\t" + std::to_string(dim) + "\t" + std::to_string(scm.size()) + "\n";
        for (size_t k = 0; k < scm.items.size(); k++)
            if (scm.alive[k])
                out += std::to_string(scm.items[k].first >> 32) + "\t" + std::to_string((uint32_t)scm.items[k].first) + "\t" +
                       python_float_str(scm.items[k].second) + "\n";
        write_binary(basepath + "_scm.mtx", out.data(), out.size());
    }
    {
        const std::string out = std::to_string(sel_off + (int64_t)last_p) + "\t" + std::to_string(last_p) + "\t" + std::to_string(last_ml) + "\n";
        write_binary(basepath + ".mdim", out.data(), out.size());
    }
    {
        std::vector<int> v;
        for (const auto &kv : gmi) v.push_back((int)kv.second);
        std::sort(v.begin(), v.end());
        write_binary(basepath + ".ixs", v.data(), v.size());
    }
}

}  // namespace host
