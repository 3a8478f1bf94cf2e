// block_pipeline.h -- one LD block through the `cusk` pipeline, shared by the `mps cusk` command (one block per
// process, /root/reference/cusk/src/cli.cpp:432-678) and by the multi-GPU block driver (many blocks per process:
// include/cusk_hip.h section 3, ci-gwas_amd/run_blocks.py).  Both callers run exactly this code, so the files of a
// sharded whole-chromosome run are byte-identical to those of per-block `mps cusk` invocations by construction.
//
// Host code against the C ABI only (no HIP headers): correlation build and both skeleton stages run on the engine's
// device, the matrix never leaves HBM between them, adjacency comes back as a bitmap, separating sets as sparse
// records, only the retained sub-matrix is gathered to the host.
#pragma once
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <chrono>
#include <cstring>
#include <memory>
#include <numeric>
#include <stdexcept>

#include "../../../include/cusk_hip.h"
#include "host_io.h"

namespace host {

constexpr int ML = CUSK_ML;

// a failed engine call: the CLI prints the message and exits with EXIT_FAILURE (gpuerrors.h:6-15), the library
// entry points return CUSK_ERR_* with the message retrievable
struct EngineError : std::runtime_error
{
    using std::runtime_error::runtime_error;
};

[[noreturn]] inline void engine_die(const char *what, cusk_engine *e)
{
    throw EngineError(std::string(what) + ": " + (e ? cusk_last_error(e) : "no engine"));
}

struct Bits
{
    int n = 0, words = 0;
    std::vector<uint64_t> w;
    bool get(int i, int j) const { return (w[(size_t)i * words + (j >> 6)] >> (j & 63)) & 1ull; }
};

inline Bits fetch_adjacency(cusk_engine *e)
{
    Bits b;
    b.n = cusk_result_n(e);
    b.words = cusk_result_words(e);
    b.w.resize((size_t)b.n * b.words);
    if (cusk_engine_download(e, b.w.data(), cusk_result_adj_bits_dev(e), b.w.size() * sizeof(uint64_t)) != CUSK_OK)
        engine_die("adjacency download", e);
    return b;
}

// parent_set.cpp:8-53: all traits, plus markers reached from a trait through marker nodes in
// at most max_depth hops.  Sorted ascending.
inline std::vector<int> subset_variables(const Bits &G, int num_var, int num_markers, int max_depth)
{
    std::vector<char> keep(num_var, 0);
    for (int i = num_markers; i < num_var; i++) keep[i] = 1;
    for (int start = num_markers; start < num_var; start++)
    {
        std::vector<char> seen(num_var, 0);
        for (int i = num_markers; i < num_var; i++) seen[i] = 1;
        std::vector<int> q{start}, nq;
        for (int depth = 0; depth < max_depth; depth++)
        {
            nq.clear();
            for (int node : q)
            {
                const uint64_t *row = &G.w[(size_t)node * G.words];
                for (int wv = 0; wv * 64 < num_markers; wv++)
                {
                    uint64_t bits = row[wv];
                    while (bits)
                    {
                        const int c = wv * 64 + __builtin_ctzll(bits);
                        bits &= bits - 1;
                        if (c < num_markers && !seen[c])
                        {
                            seen[c] = 1;
                            nq.push_back(c);
                        }
                    }
                }
            }
            q.swap(nq);
        }
        for (int i = 0; i < num_var; i++)
            if (seen[i]) keep[i] = 1;
    }
    std::vector<int> out;
    for (int i = 0; i < num_var; i++)
        if (keep[i]) out.push_back(i);
    return out;
}

// The same for max_depth == 1 from the TRAIT rows alone (`traits`: the last num_var - num_markers rows of the bitmap, row t
// = trait t): one round from every trait reaches exactly its marker neighbours.
inline std::vector<int> subset_variables_depth1(const Bits &traits, int num_var, int num_markers)
{
    std::vector<uint64_t> any((size_t)traits.words, 0ull);
    for (int t = 0; t < traits.n; t++)
        for (int w = 0; w < traits.words; w++) any[(size_t)w] |= traits.w[(size_t)t * traits.words + w];
    std::vector<int> out;
    for (int c = 0; c < num_markers; c++)
        if ((any[(size_t)(c >> 6)] >> (c & 63)) & 1ull) out.push_back(c);
    for (int i = num_markers; i < num_var; i++) out.push_back(i);
    return out;
}

// cli.cpp:561-565: how many marker-trait correlations have |atanh c| >= th0.  That is a comparison of |c| with tanh(th0):
// only the elements within 1e-6 (relative) of that value, NaN and |c| >= 1 go through the reference's expression -- its
// two double-precision logs per element cost 0.7 ms on the 200,000 correlations of a 10k-SNP x 20-trait block.
inline int count_significant(const float *mxp, size_t count, float th0)
{
    const double c_lo = std::tanh((double)th0) * (1.0 - 1e-6), c_hi = std::tanh((double)th0) * (1.0 + 1e-6);
    int num_sig = 0;
    for (size_t i = 0; i < count; i++)
    {
        const float c = mxp[i];
        const double ac = std::fabs((double)c);
        if (ac < c_lo) continue;
        if (ac > c_hi && ac < 1.0)
        {
            num_sig++;
            continue;
        }
        num_sig += (std::fabs(0.5 * (std::log(std::fabs((1 + c))) - std::log(std::fabs(1 - c)))) >= th0);
    }
    return num_sig;
}

// ReducedGC / ReducedGCS of include/mps/parent_set.h
// One separating set of the reduced result: the cell (ix, iy) of the num_var x num_var x max_level array holds s[0 .. cnt)
// followed by -1
struct SepRec
{
    int ix, iy, cnt;
    int s[ML];
};

struct Reduced
{
    size_t num_var = 0, num_phen = 0, max_level = 0;
    std::vector<int> new_to_old;
    std::vector<int> G;
    std::vector<float> C;
    std::vector<float> ess;  // cuskss
    // cusk: the separating sets, num_var^2 * max_level ints, -1 where there is none (570 retained variables: 18 MB for a
    // few thousand sets).  The single-block pipeline keeps the sets as a sorted list (sep_sparse = true) and the dense
    // array is only built for whoever asks for it (dense_sep); the .sep file is written without it (write_sep_streaming).
    mutable std::vector<int> S;
    std::vector<SepRec> sep_recs;  // ascending (ix, iy), one per cell
    bool sep_sparse = false;
    size_t num_markers() const { return num_var - num_phen; }
    bool has_sep() const { return sep_sparse || !S.empty(); }
    const std::vector<int> &dense_sep() const
    {
        if (sep_sparse && S.empty())
        {
            S.assign(num_var * num_var * max_level, -1);
            for (const SepRec &q : sep_recs)
                std::memcpy(&S[((size_t)q.ix * num_var + (size_t)q.iy) * max_level], q.s, sizeof(int) * (size_t)q.cnt);
        }
        return S;
    }
};

// arrival order -> ascending cells; records of one cell keep their order (the later one overwrites the head of the earlier,
// as in the dense reduction)
inline void sort_sep_recs(Reduced &r)
{
    const size_t k = r.num_var;
    auto less = [k](const SepRec &a, const SepRec &b) { return (size_t)a.ix * k + (size_t)a.iy < (size_t)b.ix * k + (size_t)b.iy; };
    if (!std::is_sorted(r.sep_recs.begin(), r.sep_recs.end(), less)) std::stable_sort(r.sep_recs.begin(), r.sep_recs.end(), less);
}

// The .sep file of a sparse result: the array goes out in pieces of ~1 MB that are a whole number of cells, from ONE buffer
// of -1 that is patched with the sets of the piece and restored afterwards -- no 18 MB array to fill and to read back
// from memory (the buffer stays in cache), same bytes.
inline void write_sep_streaming(const std::string &path, const Reduced &r)
{
    FILE *f = std::fopen(path.c_str(), "wb");
    if (!f) die("cannot write " + path);
    const size_t ml = r.max_level, total = r.num_var * r.num_var * ml;
    const size_t piece = ml * ((size_t)(1 << 18) / (ml ? ml : 1));  // ints
    std::vector<int> buf(std::min(piece, total), -1);
    size_t q = 0;
    for (size_t c0 = 0; c0 < total; c0 += piece)
    {
        const size_t c1 = std::min(total, c0 + piece), q0 = q;
        for (; q < r.sep_recs.size(); q++)
        {
            const SepRec &s = r.sep_recs[q];
            const size_t base = ((size_t)s.ix * r.num_var + (size_t)s.iy) * ml;
            if (base >= c1) break;
            std::memcpy(&buf[base - c0], s.s, sizeof(int) * (size_t)s.cnt);
        }
        std::fwrite(buf.data(), sizeof(int), c1 - c0, f);
        for (size_t z = q0; z < q; z++)
        {
            const SepRec &s = r.sep_recs[z];
            const size_t base = ((size_t)s.ix * r.num_var + (size_t)s.iy) * ml;
            for (int t = 0; t < s.cnt; t++) buf[base - c0 + (size_t)t] = -1;
        }
    }
    std::fclose(f);
}

inline void write_reduced(const Reduced &r, const std::string &base, bool with_sep)
{
    {
        char line[96];
        const int len = std::snprintf(line, sizeof(line), "%zu\t%zu\t%zu\n", r.num_var, r.num_phen, r.max_level);
        write_binary(base + ".mdim", line, (size_t)len);
    }
    write_binary(base + ".ixs", r.new_to_old.data(), r.new_to_old.size());
    write_binary(base + ".adj", r.G.data(), r.G.size());
    write_binary(base + ".corr", r.C.data(), r.C.size());
    if (with_sep && r.sep_sparse && r.S.empty())
        write_sep_streaming(base + ".sep", r);
    else if (with_sep)
        write_binary(base + ".sep", r.S.data(), r.S.size());
}

inline std::vector<float> gather(cusk_engine *e, const float *M_dev, int n, const std::vector<int> &P)
{
    std::vector<float> out(P.size() * P.size());
    if (cusk_gather_submatrix(e, M_dev, n, P.data(), (int)P.size(), out.data()) != CUSK_OK) engine_die("gather", e);
    return out;
}

inline std::vector<int> gather_adj(const Bits &G, const std::vector<int> &P)
{
    std::vector<int> out(P.size() * P.size());
    for (size_t a = 0; a < P.size(); a++)
        for (size_t b = 0; b < P.size(); b++) out[a * P.size() + b] = G.get(P[a], P[b]) ? 1 : 0;
    return out;
}

// parent_set.cpp:84-175 on sparse records.  Entries of a set that are not retained are dropped, the
// rest is compacted and padded with -1 to `max_level`; at most `max_level` source entries are read.
// With an index_map (stage two) the reference keys old_to_new by index_map[P[i]] although the set
// members are still in the P index space: a member that is not a key maps to 0.
inline std::vector<int> reduce_sepsets(cusk_engine *e, const std::vector<int> &P, size_t max_level,
                                       const std::vector<int> *index_map)
{
    const size_t k = P.size();
    std::vector<int> S(k * k * max_level, -1);
    const int *x = nullptr, *y = nullptr, *rs = nullptr;  // engine-owned pinned memory
    const long long cnt = cusk_result_sepsets_view(e, &x, &y, &rs);
    if (cnt < 0) engine_die("sepsets", e);
    if (cnt == 0) return S;
    std::unordered_map<int, int> pos, old_to_new;
    for (size_t i = 0; i < k; i++)
    {
        pos[P[i]] = (int)i;
        old_to_new[index_map ? (*index_map)[P[i]] : P[i]] = (int)i;
    }
    for (long long r = 0; r < cnt; r++)
    {
        auto ix = pos.find(x[r]), iy = pos.find(y[r]);
        if (ix == pos.end() || iy == pos.end()) continue;
        int *dst = &S[((size_t)ix->second * k + iy->second) * max_level];
        size_t c = 0;
        for (size_t l = 0; l < max_level && l < (size_t)ML; l++)
        {
            const int sv = rs[(size_t)r * ML + l];
            if (sv != -1 && pos.count(sv))
            {
                auto it = old_to_new.find(sv);
                dst[c++] = (it == old_to_new.end()) ? 0 : it->second;
            }
        }
    }
    return S;
}

// reduce_sepsets without the dense array: the same cells as a list in ascending (ix, iy) order.  Returns false (and leaves
// the list empty) when two records name the same cell -- the dense form's "the later record overwrites the head of the
// earlier" is then what the caller must reproduce (never seen: an ordered pair has one record).
inline bool reduce_sepsets_sparse(cusk_engine *e, const std::vector<int> &P, size_t max_level, const std::vector<int> *index_map,
                                  std::vector<SepRec> &out)
{
    out.clear();
    const size_t k = P.size();
    const int *x = nullptr, *y = nullptr, *rs = nullptr;  // engine-owned pinned memory
    const long long cnt = cusk_result_sepsets_view(e, &x, &y, &rs);
    if (cnt < 0) engine_die("sepsets", e);
    if (cnt == 0) return true;
    std::unordered_map<int, int> pos, old_to_new;
    for (size_t i = 0; i < k; i++)
    {
        pos[P[i]] = (int)i;
        old_to_new[index_map ? (*index_map)[P[i]] : P[i]] = (int)i;
    }
    out.reserve((size_t)cnt);
    for (long long r = 0; r < cnt; r++)
    {
        auto ix = pos.find(x[r]), iy = pos.find(y[r]);
        if (ix == pos.end() || iy == pos.end()) continue;
        SepRec q;
        q.ix = ix->second;
        q.iy = iy->second;
        q.cnt = 0;
        for (size_t l = 0; l < max_level && l < (size_t)ML; l++)
        {
            const int sv = rs[(size_t)r * ML + l];
            if (sv != -1 && pos.count(sv))
            {
                auto it = old_to_new.find(sv);
                q.s[q.cnt++] = (it == old_to_new.end()) ? 0 : it->second;
            }
        }
        out.push_back(q);
    }
    auto key = [k](const SepRec &q) { return (size_t)q.ix * k + (size_t)q.iy; };
    if (!std::is_sorted(out.begin(), out.end(), [&](const SepRec &a, const SepRec &b) { return key(a) < key(b); }))
        std::stable_sort(out.begin(), out.end(), [&](const SepRec &a, const SepRec &b) { return key(a) < key(b); });
    for (size_t i = 1; i < out.size(); i++)
        if (key(out[i]) == key(out[i - 1]))
        {
            out.clear();
            return false;
        }
    return true;
}

inline std::vector<int> compose(const std::vector<int> &P, const std::vector<int> *index_map)
{
    std::vector<int> out(P.size());
    for (size_t i = 0; i < P.size(); i++) out[i] = index_map ? (*index_map)[P[i]] : P[i];
    return out;
}

// device matrix that is reused from block to block (grows, never shrinks)
struct DevMat
{
    float *p = nullptr;
    size_t cap = 0;
    DevMat() = default;
    explicit DevMat(size_t count) { reserve(count); }
    DevMat(const std::vector<float> &h) { upload(h); }
    void reserve(size_t count)
    {
        if (count <= cap) return;
        cusk_dev_free(p);
        p = static_cast<float *>(cusk_dev_alloc(sizeof(float) * count));
        cap = p ? count : 0;
        if (!p) throw EngineError("device allocation of " + std::to_string(sizeof(float) * count) + " bytes failed");
    }
    void upload(const std::vector<float> &h)
    {
        reserve(h.size());
        if (cusk_dev_upload(p, h.data(), sizeof(float) * h.size()) != CUSK_OK) throw EngineError("device upload failed");
    }
    ~DevMat() { cusk_dev_free(p); }
    DevMat(const DevMat &) = delete;
    DevMat &operator=(const DevMat &) = delete;
};

// read-only memory map of the .bed (a whole-genome file is read block by block, by whichever rank owns the block)
struct MappedFile
{
    const unsigned char *data = nullptr;
    size_t size = 0;
    int fd = -1;
    void open(const std::string &path)
    {
        close();
        fd = ::open(path.c_str(), O_RDONLY);
        if (fd < 0) die("file or directory not found: " + path);
        struct stat st;
        if (fstat(fd, &st) != 0) die("cannot stat " + path);
        size = (size_t)st.st_size;
        if (size)
        {
            void *p = mmap(nullptr, size, PROT_READ, MAP_PRIVATE, fd, 0);
            if (p == MAP_FAILED) die("cannot map " + path);
            data = static_cast<const unsigned char *>(p);
        }
    }
    void close()
    {
        if (data) munmap(const_cast<unsigned char *>(data), size);
        if (fd >= 0) ::close(fd);
        data = nullptr;
        size = 0;
        fd = -1;
    }
    ~MappedFile() { close(); }
    MappedFile() = default;
    MappedFile(const MappedFile &) = delete;
    MappedFile &operator=(const MappedFile &) = delete;
};

// what `mps cusk` loads before it turns to its block (cli.cpp:458-497); loaded once, shared by all blocks
struct CuskInputs
{
    std::string phen_path, bfiles, block_path;
    float alpha = 0.0f;
    int max_level = 0, max_level_two = 0, depth = 1;
    std::string full_corrmats_dir;  // not empty: also write <stem>.all_corrs there (cli.cpp:651-658)
    Phen phen;
    BedDims dims;
    BimInfo bim;
    std::vector<Block> blocks;
    float Th[ML + 1];
    MappedFile bed;
    // all markers' means / stds, read once when several blocks are run from one process (empty: line-range reads)
    std::vector<float> means_all, stds_all;

    // cli.cpp:458-497 (path checks, .phen, .dim, .bim, .blocks with the bounds check, thresholds)
    void load(std::ostream *log)
    {
        if (log) *log << "Checking paths" << std::endl;
        for (const char *sfx : {".bed", ".dim", ".means", ".stds", ".bim"}) check_path(bfiles + sfx);
        if (!bed_has_valid_magic(bfiles + ".bed")) die("unexpected magic number in bed file.");
        check_path(phen_path);
        check_path(block_path);
        phen = load_phen(phen_path);
        dims = read_dims(bfiles + ".dim");
        if (phen.num_samples != dims.num_samples) die("different num samples in phen and dims");
        bim = read_bim(bfiles + ".bim");
        if (log) *log << "Found " << phen.num_phen << " phenotypes" << std::endl;
        if (log) *log << "Loading blocks" << std::endl;
        blocks = read_blocks(block_path);
        if (log) *log << "Found " << blocks.size() << " blocks" << std::endl;
        for (const Block &b : blocks)
            if (b.first >= bim.markers_on(b.chr) || b.last >= bim.markers_on(b.chr))
                die("block out of bounds with first_ix: " + std::to_string(b.first) + " last_ix: " + std::to_string(b.last));
        cusk_threshold_array((int)dims.num_samples, alpha, Th);
        bed.open(bfiles + ".bed");
    }
    void load_all_marker_stats()
    {
        means_all = read_floats_line_range(bfiles + ".means", 0, std::numeric_limits<size_t>::max());
        stds_all = read_floats_line_range(bfiles + ".stds", 0, std::numeric_limits<size_t>::max());
    }
    size_t first_marker(const Block &b) const { return bim.start_of(b.chr) + b.first; }
};

// the inputs all blocks share, resident on one device (cusk_blockset_stage): marker g's genotypes start at
// bed + g * bytes_per_col, means / stds are indexed by global marker
struct StagedInputs
{
    unsigned char *bed = nullptr;
    float *phen = nullptr, *means = nullptr, *stds = nullptr;
    void release()
    {
        cusk_dev_free(bed);
        cusk_dev_free(phen);
        cusk_dev_free(means);
        cusk_dev_free(stds);
        bed = nullptr;
        phen = means = stds = nullptr;
    }
};

struct BlockStats
{
    int skipped = 0;         // cli.cpp:572-576: no marginally significant marker-trait correlation, no output
    int num_sig = 0;
    long long markers = 0;   // block size
    long long retained = 0;  // markers in the written result
    long long tests[2] = {0, 0};  // CI tests of stage one / stage two (SURVEY.md 8d definition)
    cusk_stats stage[2];
    // wall-clock phases, ms: inputs (bed slice, means, stds), correlation build, stage one, prune + gather,
    // stage two, reduction
    double ms_inputs = 0, ms_corr = 0, ms_stage1 = 0, ms_prune = 0, ms_stage2 = 0, ms_reduce = 0;
};

// per-caller device scratch that survives from block to block
struct BlockScratch
{
    DevMat C, C2;
    // the NEXT block's correlation matrix, being built on the engine's third stream while this block is swept
    // (cusk_corr_build_begin / _end): pending >= 0 = its block index, pending_m = its marker count
    DevMat Cnext;
    int pending = -1;
    size_t pending_m = 0;
    void swap_next()
    {
        std::swap(C.p, Cnext.p);
        std::swap(C.cap, Cnext.cap);
    }
};

// cli.cpp:521-677 for blocks[block_index] on the engine's device.  Returns false when the block is skipped.
// next_index >= 0 (block driver, device-resident inputs): the correlation build of that block is started as soon as this
// block's own matrix is complete and runs beside this block's sweeps; the call for block next_index then only waits for it.
inline bool run_cusk_block(cusk_engine *e, const CuskInputs &in, int block_index, BlockScratch &scr, Reduced &out,
                           std::string &stem, BlockStats &bs, std::ostream *log, const StagedInputs *staged = nullptr,
                           int next_index = -1)
{
    using clk = std::chrono::steady_clock;
    auto ms_since = [](clk::time_point &t) {
        const auto now = clk::now();
        const double v = std::chrono::duration<double, std::milli>(now - t).count();
        t = now;
        return v;
    };
    if (block_index < 0 || (size_t)block_index >= in.blocks.size()) die("block index out of range");
    if (cusk_engine_bind_thread(e) != CUSK_OK) engine_die("bind thread", e);  // the scratch matrices go to e's device
    const Block &block = in.blocks[block_index];
    stem = block.file_stem();
    const size_t m = block.size(), N = in.dims.num_samples, p = in.phen.num_phen;
    bs = BlockStats();
    bs.markers = (long long)m;
    auto t = clk::now();
    if (log)
    {
        *log << "\nProcessing block " << block_index + 1 << " / " << in.blocks.size() << std::endl;
        *log << "Block size: " << m << std::endl;
        *log << "Loading bed data" << std::endl;
    }
    const size_t g0 = in.first_marker(block), g1 = g0 + m - 1;
    const size_t bpc = in.dims.bytes_per_col();
    if (3 + (g1 + 1) * bpc > in.bed.size) die("bed file is shorter than .dim / .bim say");
    const unsigned char *bed = in.bed.data + 3 + g0 * bpc;  // io.cpp:238-249
    const float *phen = in.phen.data.data();
    std::vector<float> means_v, stds_v;
    const float *means, *stds;
    if (staged)
    {  // device-resident inputs: no per-block PCIe traffic
        bed = staged->bed + g0 * bpc;
        phen = staged->phen;
        means = staged->means + g0;
        stds = staged->stds + g0;
    }
    else if (!in.means_all.empty())
    {
        if (g1 >= in.means_all.size() || g1 >= in.stds_all.size()) die("block size and number of means or stds differ");
        means = in.means_all.data() + g0;
        stds = in.stds_all.data() + g0;
    }
    else
    {
        means_v = read_floats_line_range(in.bfiles + ".means", g0, g1);
        stds_v = read_floats_line_range(in.bfiles + ".stds", g0, g1);
        if (means_v.size() != m || stds_v.size() != m) die("block size and number of means or stds differ");
        means = means_v.data();
        stds = stds_v.data();
    }
    bs.ms_inputs = ms_since(t);

    const size_t n = m + p;
    if (log)
    {
        *log << "Checking for significant marker - phen correlations" << std::endl;
        *log << "Computing all correlations" << std::endl;
    }
    std::vector<float> mxp(m * p);
    if (scr.pending == block_index && scr.pending_m == m)
    {  // built ahead, beside the previous block's sweeps
        scr.pending = -1;
        scr.swap_next();
        if (cusk_corr_build_end(e, mxp.data()) != CUSK_OK) engine_die("correlation build (end)", e);
    }
    else
    {
        if (scr.pending >= 0)
        {  // a build for another block is in flight (the caller changed its mind): let it finish, drop it
            scr.pending = -1;
            if (cusk_corr_build_end(e, nullptr) != CUSK_OK) engine_die("correlation build (end)", e);
        }
        scr.C.reserve(n * n);
        if (cusk_corr_build(e, bed, phen, m, N, p, means, stds, scr.C.p, mxp.data()) != CUSK_OK)
            engine_die("correlation build", e);
    }
    if (staged && p > 0 && next_index >= 0 && (size_t)next_index < in.blocks.size() && next_index != block_index)
    {
        const Block &nb = in.blocks[next_index];
        const size_t m2 = nb.size(), h0 = in.first_marker(nb), n2 = m2 + p;
        if (m2 > 0 && 3 + (h0 + m2) * bpc <= in.bed.size)
        {
            scr.Cnext.reserve(n2 * n2);
            if (cusk_corr_build_begin(e, staged->bed + h0 * bpc, staged->phen, m2, N, p, staged->means + h0, staged->stds + h0,
                                      scr.Cnext.p) != CUSK_OK)
                engine_die("correlation build (begin)", e);
            scr.pending = next_index;
            scr.pending_m = m2;
        }
    }
    bs.ms_corr = ms_since(t);
    // cli.cpp:561-576: blocks without any marginally significant marker-trait correlation are skipped
    const int num_sig = count_significant(mxp.data(), mxp.size(), in.Th[0]);
    bs.num_sig = num_sig;
    if (num_sig > 0)
    {
        if (log) *log << "Found " << num_sig << " marker - phen correlations. Proceeding." << std::endl;
    }
    else
    {
        if (log) *log << "No significant correlations found. Skipping block." << std::endl;
        bs.skipped = 1;
        return false;
    }
    if (!in.full_corrmats_dir.empty())
    {  // cli.cpp:27,651-658 (compile-time switch WRITE_FULL_CORRMATS in the reference)
        std::vector<float> full(n * n);
        if (cusk_dev_download(full.data(), scr.C.p, sizeof(float) * n * n) != CUSK_OK) engine_die("correlation matrix download", e);
        write_binary(make_path(in.full_corrmats_dir, stem, ".all_corrs"), full.data(), full.size());
    }

    if (log) *log << "Running cuPC" << std::endl;
    cusk_engine_set_option(e, "assume_symmetric", 1);  // cusk_corr_build mirrors every element
    cusk_stats &st = bs.stage[0];
    if (cusk_run_skeleton(e, scr.C.p, (int)n, in.Th, in.max_level, &st) != CUSK_OK) engine_die("Skeleton", e);
    bs.ms_stage1 = ms_since(t);
    for (int l = 0; l < st.levels_run; l++)
    {
        bs.tests[0] += st.tests[l];
        if (log)
            *log << "level " << l << ": max degree " << st.max_degree[l] << ", " << st.tests[l] << " tests, "
                 << st.level_ms[l] * 1e-3 << " s" << std::endl;
    }
    std::vector<int> P;
    if (in.depth == 1 && p > 0)
    {  // only the trait rows travel (25 KB of the 12.6 MB bitmap of a 10k block)
        Bits T;
        T.n = (int)p;
        T.words = cusk_result_words(e);
        T.w.resize((size_t)T.n * T.words);
        if (cusk_result_adj_rows(e, (int)m, (int)p, T.w.data()) != CUSK_OK) engine_die("adjacency download", e);
        P = subset_variables_depth1(T, (int)n, (int)m);
    }
    else
    {
        Bits G = fetch_adjacency(e);
        P = subset_variables(G, (int)n, (int)m, in.depth);
    }
    Reduced gcs;
    gcs.num_var = P.size();
    gcs.num_phen = p;
    gcs.max_level = (size_t)in.max_level;
    gcs.new_to_old = P;
    // cli.cpp:62-87: the retained sub-matrix is the input of stage two -- gathered device to device
    const int k = (int)gcs.num_var;
    scr.C2.reserve((size_t)k * k);
    if (cusk_gather_submatrix_dev(e, scr.C.p, (int)n, P.data(), k, scr.C2.p) != CUSK_OK) engine_die("gather", e);
    bs.ms_prune = ms_since(t);
    // (the stage-one separating sets of cli.cpp:673 are never read again: stage two recomputes them)

    if (log) *log << "Starting second cusk stage" << std::endl;
    // Skeleton again on the reduced set, starting from the complete graph
    cusk_engine_set_option(e, "assume_symmetric", 0);
    cusk_stats &st2 = bs.stage[1];
    if (cusk_run_skeleton(e, scr.C2.p, k, in.Th, in.max_level_two, &st2) != CUSK_OK) engine_die("Skeleton (stage two)", e);
    bs.ms_stage2 = ms_since(t);
    for (int l = 0; l < st2.levels_run; l++) bs.tests[1] += st2.tests[l];
    Bits G2 = fetch_adjacency(e);
    std::vector<int> P2 = subset_variables(G2, k, (int)gcs.num_markers(), in.depth);
    out = Reduced();
    out.num_var = P2.size();
    out.num_phen = p;
    out.max_level = ML;
    out.new_to_old = compose(P2, &gcs.new_to_old);
    out.G = gather_adj(G2, P2);
    out.C = gather(e, scr.C2.p, k, P2);
    // (the sets as a list; the 18 MB dense array only for callers that ask for it, the .sep file is streamed)
    out.sep_sparse = reduce_sepsets_sparse(e, P2, ML, &gcs.new_to_old, out.sep_recs);
    if (!out.sep_sparse) out.S = reduce_sepsets(e, P2, ML, &gcs.new_to_old);
    bs.retained = (long long)out.num_markers();
    if (log) *log << "Retained " << out.num_markers() << " markers" << std::endl;
    bs.ms_reduce = ms_since(t);
    return true;
}

}  // namespace host
