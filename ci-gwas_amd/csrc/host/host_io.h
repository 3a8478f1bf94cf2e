// host_io.h -- input loaders and result writers of the `mps cusk` / `mps cuskss` host program.
//
// Own implementation of the on-disk formats the reference reads and writes for this path
// (/root/reference/cusk/src/io.cpp, bim.cpp, phen.cpp,
// marker_summary_stats.cpp, marker_trait_summary_stats.cpp, trait_summary_stats.cpp,
// include/mps/parent_set.h:42-52,99-108).  Text parsing is buffered and binary output is
// written with one fwrite per array (the reference issues one 4-byte write per value).
#pragma once
#include <fcntl.h>
#include <unistd.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <limits>
#include <sstream>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <vector>

namespace host {

// Bad input ends the reference's program with a message and status 1 (cli.cpp:185-192, io.cpp:399-406).  Here it is
// an exception so that the same loaders serve the library entry points: `mps` catches it in main() and does exactly
// that, the cusk_blockset_* calls turn it into an error code.
struct Fatal : std::runtime_error
{
    using std::runtime_error::runtime_error;
};
[[noreturn]] inline void die(const std::string &msg) { throw Fatal(msg); }

inline bool path_exists(const std::string &p)
{
    std::ifstream f(p);
    return f.good();
}

// cli.cpp check_path: missing inputs end the program with status 1
inline void check_path(const std::string &p)
{
    if (!path_exists(p)) die("file or directory not found: " + p);
}

inline std::vector<std::string> split_ws(const std::string &line)
{
    std::vector<std::string> out;
    std::istringstream ss(line);
    std::string w;
    while (ss >> w) out.push_back(w);
    return out;
}

inline std::string make_path(const std::string &dir, const std::string &stem, const std::string &suffix)
{
    if (dir.empty() || dir.back() == '/') return dir + stem + suffix;
    return dir + "/" + stem + suffix;
}

struct Block
{
    std::string chr;
    size_t first = 0, last = 0, global_offset = 0;  // first/last: 0-based inclusive within the chromosome
    size_t size() const { return last - first + 1; }
    size_t first_global() const { return first + global_offset; }
    size_t last_global() const { return last + global_offset; }
    std::string file_stem() const { return chr + "_" + std::to_string(first) + "_" + std::to_string(last); }
};

// io.cpp:74-101: the global offset of a chromosome accumulates the sizes of the blocks listed before it
inline std::vector<Block> read_blocks(const std::string &path)
{
    std::ifstream f(path);
    std::vector<Block> blocks;
    std::string line, cur;
    size_t offset = 0, on_chr = 0;
    while (std::getline(f, line))
    {
        auto w = split_ws(line);
        if (w.size() < 3) continue;
        if (w[0] != cur)
        {
            cur = w[0];
            offset += on_chr;
            on_chr = 0;
        }
        Block b;
        b.chr = w[0];
        b.first = std::stoul(w[1]);
        b.last = std::stoul(w[2]);
        b.global_offset = offset;
        blocks.push_back(b);
        on_chr += b.size();
    }
    return blocks;
}

// <stem>.dim: "num_samples\tnum_markers" (io.h:30-39)
struct BedDims
{
    size_t num_samples = 0, num_markers = 0;
    size_t bytes_per_col() const { return (num_samples + 3) / 4; }
};
// io.cpp count_lines: number of newline-terminated lines (a last line without newline counts too)
inline size_t count_lines(const std::string &path)
{
    std::ifstream f(path);
    std::string line;
    size_t k = 0;
    while (std::getline(f, line)) ++k;
    return k;
}
inline BedDims read_dims(const std::string &path)
{
    std::ifstream f(path);
    std::string line;
    std::getline(f, line);
    auto w = split_ws(line);
    if (w.size() < 2) die("bad .dim file: " + path);
    BedDims d;
    d.num_samples = std::stoul(w[0]);
    d.num_markers = std::stoul(w[1]);
    return d;
}

// .bim: only column 1 (chromosome id) and the line order matter (bim.cpp:20-48)
struct BimInfo
{
    std::vector<std::string> chr_ids;
    std::vector<size_t> num_on_chr, chr_start;
    std::unordered_map<std::string, size_t> ix;
    size_t num_lines = 0;
    size_t chr_index(const std::string &c) const
    {
        auto it = ix.find(c);
        if (it == ix.end()) die("chr id does not match .bim content");
        return it->second;
    }
    size_t markers_on(const std::string &c) const { return num_on_chr[chr_index(c)]; }
    size_t start_of(const std::string &c) const { return chr_start[chr_index(c)]; }
};
inline BimInfo read_bim(const std::string &path)
{
    BimInfo b;
    std::ifstream f(path);
    std::string line;
    while (std::getline(f, line))
    {
        std::istringstream ss(line);
        std::string chr;
        ss >> chr;
        if (b.num_lines == 0 || chr != b.chr_ids.back())
        {
            b.chr_start.push_back(b.num_lines);
            b.ix[chr] = b.chr_ids.size();
            b.chr_ids.push_back(chr);
            b.num_on_chr.push_back(0);
        }
        ++b.num_on_chr.back();
        ++b.num_lines;
    }
    return b;
}

// SNP-major .bed after the 3 magic bytes 6c 1b 01 (io.cpp:238-249)
inline std::vector<unsigned char> read_bed_block(const std::string &path, const Block &blk, const BedDims &dims,
                                                 const BimInfo &bim)
{
    const size_t start = bim.start_of(blk.chr) + blk.first;
    std::vector<unsigned char> out(dims.bytes_per_col() * blk.size(), 0);
    std::ifstream f(path, std::ios::binary);
    f.seekg(3 + (std::streamoff)(start * dims.bytes_per_col()));
    f.read(reinterpret_cast<char *>(out.data()), (std::streamsize)out.size());
    return out;
}

inline bool bed_has_valid_magic(const std::string &path)
{
    unsigned char m[3] = {0, 0, 0};
    std::ifstream f(path, std::ios::binary);
    f.read(reinterpret_cast<char *>(m), 3);
    return m[0] == 0x6c && m[1] == 0x1b && m[2] == 0x01;
}

// one float per line, lines first..last inclusive (io.cpp:137-158)
inline std::vector<float> read_floats_line_range(const std::string &path, size_t first, size_t last)
{
    std::ifstream f(path);
    std::vector<float> out;
    std::string line;
    size_t i = 0;
    while (std::getline(f, line))
    {
        if (i > last) break;
        if (i >= first) out.push_back(std::stof(line));
        ++i;
    }
    return out;
}

inline std::vector<int> read_ints_lines(const std::string &path)
{
    std::ifstream f(path);
    std::vector<int> out;
    std::string line;
    while (std::getline(f, line))
        if (!line.empty()) out.push_back(std::stoi(line));
    return out;
}

template <typename T>
inline std::vector<T> read_binary(const std::string &path)
{
    std::ifstream f(path, std::ios::binary | std::ios::ate);
    const std::streamsize bytes = f.tellg();
    std::vector<T> out((size_t)std::max<std::streamsize>(bytes, 0) / sizeof(T));
    f.seekg(0);
    f.read(reinterpret_cast<char *>(out.data()), (std::streamsize)(out.size() * sizeof(T)));
    return out;
}

template <typename T>
inline void write_binary(const std::string &path, const T *data, size_t count)
{
    // (plain descriptors: a job writes five small files per block, and a stdio stream costs a buffer and an fstat each)
    const int fd = ::open(path.c_str(), O_WRONLY | O_CREAT | O_TRUNC, 0666);
    if (fd < 0) die("cannot write " + path);
    const char *p = reinterpret_cast<const char *>(data);
    size_t left = sizeof(T) * count;
    while (left)
    {
        const ssize_t w = ::write(fd, p, left);
        if (w < 0)
        {
            ::close(fd);
            die("cannot write " + path);
        }
        p += w;
        left -= (size_t)w;
    }
    ::close(fd);
}

// .phen: header skipped, FID IID dropped, "NA" -> NaN, returned column-major (phen.cpp:9-74)
struct Phen
{
    size_t num_samples = 0, num_phen = 0;
    std::vector<float> data;
};
inline Phen load_phen(const std::string &path)
{
    Phen p;
    std::ifstream f(path);
    std::string line;
    std::getline(f, line);
    std::vector<float> rows;
    while (std::getline(f, line))
    {
        auto w = split_ws(line);
        if (w.size() < 2) continue;
        const size_t np = w.size() - 2;
        if (p.num_samples == 0)
            p.num_phen = np;
        else if (np != p.num_phen)
            die("Inconsistent row width in .phen file \nLine: " + line);
        for (size_t i = 2; i < w.size(); i++)
            rows.push_back(w[i] == "NA" ? std::numeric_limits<float>::quiet_NaN() : (float)std::atof(w[i].c_str()));
        ++p.num_samples;
    }
    p.data.resize(rows.size());
    for (size_t k = 0; k < p.num_phen; k++)
        for (size_t i = 0; i < p.num_samples; i++) p.data[k * p.num_samples + i] = rows[p.num_phen * i + k];
    return p;
}

// mxm: raw f32, lower triangle incl. diagonal, row-major; NaN -> 0 (marker_summary_stats.cpp:8-24).
// Written straight into the top-left m x m corner of an n x n row-major matrix.
inline size_t mxm_num_markers(const std::string &path)
{
    std::ifstream f(path, std::ios::binary | std::ios::ate);
    const size_t cnt = (size_t)f.tellg() / sizeof(float);
    return (size_t)((std::sqrt(8.0 * (double)cnt + 1.0) - 1.0) / 2.0);
}
inline void load_mxm_into(const std::string &path, size_t m, float *sq, size_t n)
{
    std::vector<float> tri = read_binary<float>(path);
    size_t k = 0;
    for (size_t i = 0; i < m; i++)
        for (size_t j = 0; j <= i; j++)
        {
            float v = tri[k++];
            if (std::isnan(v)) v = 0.0f;
            sq[i * n + j] = v;
            sq[j * n + i] = v;
        }
}

inline bool is_na_token(const std::string &s, bool with_upper)
{
    return s == "NA" || s == "NaN" || s == "nan" || (with_upper && s == "NAN");
}

// ESS from a standard error: ((1 - r^2) / se)^2 with the reference's float/double mix
// (marker_trait_summary_stats.cpp:161-164, trait_summary_stats.cpp:150-152)
inline float ess_from_se(float r, float se)
{
    const float ss_sqrt = (float)((1.0 - (double)(r * r)) / (double)se);
    return ss_sqrt * ss_sqrt;
}

// mxp (+ se): text, header "chr snp ref <traits...>", one row per marker genome-wide; rows are
// selected by ascending global line index (marker_trait_summary_stats.cpp:40-299)
struct Mxp
{
    size_t num_markers = 0, num_phen = 0;
    std::vector<float> corr, ess;  // row-major num_markers x num_phen; ess only with se
};
inline Mxp load_mxp(const std::string &path, const std::string &se_path, const std::vector<size_t> &rows)
{
    Mxp out;
    std::ifstream f(path);
    std::string line, sline;
    if (!std::getline(f, line)) die("marker-trait summary stat file seems to be empty");
    auto header = split_ws(line);
    if (header.size() < 3 || header[0] != "chr" || header[1] != "snp" || header[2] != "ref")
        die("marker-trait summary stat file has bad header");
    out.num_phen = header.size() - 3;
    const bool het = !se_path.empty();
    std::ifstream fs;
    if (het)
    {
        fs.open(se_path);
        if (!std::getline(fs, sline)) die("marker-trait se file seems to be empty");
    }
    size_t ln = 0, next = 0;
    while (next < rows.size() && std::getline(f, line))
    {
        if (het) std::getline(fs, sline);
        if (ln == rows[next])
        {
            auto w = split_ws(line);
            std::vector<std::string> ws;
            if (het) ws = split_ws(sline);
            for (size_t j = 3; j < out.num_phen + 3; j++)
            {
                if (is_na_token(w[j], het))
                {
                    out.corr.push_back(0.0f);
                    if (het) out.ess.push_back(std::numeric_limits<float>::quiet_NaN());
                }
                else
                {
                    const float r = std::stof(w[j]);
                    out.corr.push_back(r);
                    if (het) out.ess.push_back(ess_from_se(r, std::stof(ws[j])));
                }
            }
            ++out.num_markers;
            ++next;
        }
        ++ln;
    }
    return out;
}

// pxp (+ se): header = trait names, each row "name v1..vp", upper triangle mirrored
// (trait_summary_stats.cpp:5-169)
struct Pxp
{
    size_t num_phen = 0;
    std::vector<float> corr, ess;  // p x p
};
inline Pxp load_pxp(const std::string &path, const std::string &se_path, float sample_size)
{
    Pxp out;
    std::ifstream f(path);
    std::string line, sline;
    if (!std::getline(f, line)) die("trait summary stat file seems to be empty");
    const size_t p = split_ws(line).size();
    out.num_phen = p;
    const bool het = !se_path.empty();
    out.corr.assign(p * p, 1.0f);
    out.ess.assign(p * p, het ? 0.0f : sample_size);
    std::ifstream fs;
    if (het)
    {
        fs.open(se_path);
        if (!std::getline(fs, sline)) die("trait summary se file seems to be empty");
    }
    size_t row = 0;
    while (std::getline(f, line))
    {
        if (het) std::getline(fs, sline);
        auto w = split_ws(line);
        if (w.empty()) break;
        std::vector<std::string> ws;
        if (het) ws = split_ws(sline);
        for (size_t j = 1; j <= p && row < p; j++)
        {
            float r = std::stof(w[j]);
            if (het)
            {
                if (std::isnan(r))
                {
                    out.corr[row * p + j - 1] = 0.0f;
                    out.ess[row * p + j - 1] = std::numeric_limits<float>::quiet_NaN();
                }
                else
                {
                    out.corr[row * p + j - 1] = r;
                    out.ess[row * p + j - 1] = ess_from_se(r, std::stof(ws[j]));
                }
            }
            else
            {
                out.corr[row * p + j - 1] = std::isnan(r) ? 0.0f : r;
            }
        }
        ++row;
    }
    for (size_t i = 0; i < p; i++)
        for (size_t j = i + 1; j < p; j++)
        {
            out.corr[j * p + i] = out.corr[i * p + j];
            if (het) out.ess[j * p + i] = out.ess[i * p + j];
        }
    return out;
}

}  // namespace host
