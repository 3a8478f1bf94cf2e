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
#include <thread>
#include <unordered_map>
#include <vector>

#include "block_pipeline.h"

namespace host {

// f"{numpy.float32(v)}" as the reference's write_mm prints it (merge_blocks.py:309-318): numpy scalars format through
// Python's float, i.e. repr(float(v)) -- the shortest digits that round-trip the DOUBLE value, positional when the decimal
// exponent lies in [-4, 16), else scientific with at least two exponent digits
inline void append_python_float(std::string &out, float v)
{
    if (v != v)
    {
        out += "nan";
        return;
    }
    if (v == std::numeric_limits<float>::infinity() || v == -std::numeric_limits<float>::infinity())
    {
        out += (v > 0 ? "inf" : "-inf");
        return;
    }
    if (v == 0.0f)
    {
        out += std::signbit(v) ? "-0.0" : "0.0";
        return;
    }
    char buf[40];
    const auto r = std::to_chars(buf, buf + sizeof(buf) - 1, (double)v, std::chars_format::scientific);  // shortest round-trip digits
    *r.ptr = 0;
    const char *p = buf, *end = r.ptr;
    if (*p == '-')
    {
        out += '-';
        p++;
    }
    const char *e = end;
    while (e > p && *(e - 1) != 'e') e--;  // e points behind the 'e'
    const int e10 = std::atoi(e);
    const char *mend = e - 1;  // mantissa = [p, mend): d or d.ddd
    if (e10 < -4 || e10 >= 16)
    {  // (to_chars writes the exponent as Python does: sign, at least two digits)
        out.append(p, (size_t)(end - p));
        return;
    }
    char digits[24];
    int nd = 0;
    for (const char *q = p; q < mend; q++)
        if (*q != '.') digits[nd++] = *q;
    if (e10 >= 0)
    {
        const int ip = e10 + 1;  // digits before the point
        if (nd <= ip)
        {
            out.append(digits, (size_t)nd);
            out.append((size_t)(ip - nd), '0');
            out += ".0";
        }
        else
        {
            out.append(digits, (size_t)ip);
            out += '.';
            out.append(digits + ip, (size_t)(nd - ip));
        }
    }
    else
    {
        out += "0.";
        out.append((size_t)(-e10 - 1), '0');
        out.append(digits, (size_t)nd);
    }
}

// Insertion-ordered (row, column) -> value store with the update / delete semantics of a Python dict.  Keys that involve a
// marker are unique to their block (merged marker indices grow from block to block), so only the trait x trait keys can
// be hit twice: those are found through a small dense table, everything else is appended without a look-up (a hash map
// over all 40,000 correlation entries of a 25-block batch cost 10 ms).
template <typename V>
struct OrderedEntries
{
    struct Item
    {
        int64_t i, j;
        V v;
        char alive;
    };
    std::vector<Item> items;
    int64_t np1 = 0;            // keys with i < np1 and j < np1 go through the table
    std::vector<long long> at;  // np1 x np1: position in items, -1 = absent
    size_t live = 0;
    void init(int64_t num_phen)
    {
        np1 = num_phen + 1;
        at.assign((size_t)(np1 * np1), -1);
    }
    bool small(int64_t i, int64_t j) const { return i >= 0 && j >= 0 && i < np1 && j < np1; }
    void put(int64_t i, int64_t j, V v)
    {
        if (small(i, j))
        {
            long long &p = at[(size_t)(i * np1 + j)];
            if (p >= 0)
            {
                items[(size_t)p].v = v;
                return;
            }
            p = (long long)items.size();
        }
        items.push_back(Item{i, j, v, 1});
        live++;
    }
    bool has_small(int64_t i, int64_t j) const { return small(i, j) && at[(size_t)(i * np1 + j)] >= 0; }
    void drop_small(int64_t i, int64_t j)
    {
        long long &p = at[(size_t)(i * np1 + j)];
        if (p < 0) return;
        items[(size_t)p].alive = 0;
        p = -1;
        live--;
    }
};

struct MergeInput
{
    bool present = false;
    size_t num_var = 0, num_phen = 0, max_level = 0;
    const int *ixs = nullptr, *adj = nullptr;
    const float *corr = nullptr;
};

inline void append_int(std::string &out, long long v)
{
    char buf[24];
    auto r = std::to_chars(buf, buf + sizeof(buf), v);
    out.append(buf, r.ptr);
}

// blocks: one entry per line of the .blocks file, in file order (absent: the block wrote no files);
// block_sizes: markers of every listed block.  Writes <basepath>_sam.mtx, _scm.mtx, .mdim, .ixs.
inline void merge_blocks_to_files(const std::vector<MergeInput> &blocks, const std::vector<size_t> &block_sizes, const std::string &basepath)
{
    OrderedEntries<int> sam;
    OrderedEntries<float> scm;
    std::vector<int> gmi;  // global (.bim row) index of every selected marker (merged marker indices are unique: a plain list)
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
        if (!any)
        {
            sam.init(np_);
            scm.init(np_);
        }
        else if ((size_t)np_ != last_p)
            die("blocks with different numbers of traits cannot be merged");
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
            // :336-341 (0-based i, j < num_p against 1-based keys: links of the last trait are never intersected); the
            // block's own keys with i, j < num_p are its adjacency among the traits 1 .. num_p - 1
            for (int64_t i = 1; i < np_; i++)
                for (int64_t j = 1; j < np_; j++)
                    if (sam.has_small(i, j) && b.adj[(nm + i - 1) * nv + (nm + j - 1)] == 0) sam.drop_small(i, j);
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
        for (int64_t d = 0; d < nm; d++) gmi.push_back(b.ixs[d] + (int)glob_off);  // :54-72
        sel_off += nm;
        glob_off += (int64_t)block_sizes[index];
        last_p = b.num_phen;
        last_ml = b.max_level;
        any = true;
    }
    if (!any) die("no block output to merge");
    const bool prof = std::getenv("CUSK_BATCH_PROF") != nullptr;
    const auto tp0 = std::chrono::steady_clock::now();
    int64_t dim = 0;  // :300-303: both headers carry the adjacency's largest row index
    for (const auto &it : sam.items)
        if (it.alive) dim = std::max<int64_t>(dim, it.i);
    {
        std::string out = "%%MatrixMarket matrix coordinate integer general\n";
        out.reserve(64 + sam.live * 24);
        append_int(out, dim), out += '\t', append_int(out, dim), out += '\t', append_int(out, (long long)sam.live), out += '\n';
        for (const auto &it : sam.items)
            if (it.alive) append_int(out, it.i), out += '\t', append_int(out, it.j), out += '\t', append_int(out, it.v), out += '\n';
        write_binary(basepath + "_sam.mtx", out.data(), out.size());
    }
    {
        // (tens of thousands of shortest-digit conversions: a few threads format consecutive ranges of the entries)
        std::string head = "%%MatrixMarket matrix coordinate real general\n";
        append_int(head, dim), head += '\t', append_int(head, dim), head += '\t', append_int(head, (long long)scm.live), head += '\n';
        const size_t ni = scm.items.size();
        const unsigned nt = (unsigned)std::min<size_t>(16, std::max<size_t>(1, ni / 2048));
        std::vector<std::string> part(nt);
        auto work = [&](unsigned t) {
            const size_t a = ni * t / nt, b = ni * (t + 1) / nt;
            std::string &out = part[t];
            out.reserve((b - a) * 36);
            for (size_t k = a; k < b; k++)
            {
                const auto &it = scm.items[k];
                if (!it.alive) continue;
                append_int(out, it.i), out += '\t', append_int(out, it.j), out += '\t';
                append_python_float(out, it.v);
                out += '\n';
            }
        };
        if (nt == 1)
            work(0);
        else
        {
            std::vector<std::thread> th;
            for (unsigned t = 0; t < nt; t++) th.emplace_back(work, t);
            for (auto &x : th) x.join();
        }
        const auto tp1 = std::chrono::steady_clock::now();
        std::string out = std::move(head);
        for (const std::string &q : part) out += q;
        write_binary(basepath + "_scm.mtx", out.data(), out.size());
        if (prof)
            std::fprintf(stderr, "[mergeprof] %zu scm entries: format %.0f us, concat + write %.0f us\n", scm.live,
                         std::chrono::duration<double, std::micro>(tp1 - tp0).count(),
                         std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - tp1).count());
    }
    {
        std::string out;
        append_int(out, sel_off + (long long)last_p), out += '\t', append_int(out, (long long)last_p), out += '\t', append_int(out, (long long)last_ml), out += '\n';
        write_binary(basepath + ".mdim", out.data(), out.size());
    }
    std::sort(gmi.begin(), gmi.end());
    write_binary(basepath + ".ixs", gmi.data(), gmi.size());
}

}  // namespace host
