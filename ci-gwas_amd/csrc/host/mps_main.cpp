// mps_main.cpp -- `mps`-compatible host program for the cusk path on MI355X.
//
// Same positional argv contracts, stdout milestones, exit codes and output files as the
// reference's native CLI (/root/reference/cusk/apps/mps.cpp:17-121, src/cli.cpp:194-346,
// :432-678), so that ci-gwas.py (or this repo's cli shim) can call it unchanged:
//   mps cusk   <.phen> <bfiles> <.blocks> <alpha> <max-level> <max-level-two> <depth> <outdir> <block-index>
//   mps cuskss <mxm> <mxp> <mxp_se> <pxp> <pxp_se> <time_index> <block_index> <blockfile>
//              <marker_indices> <alpha> <l1> <l2> <depth> <num_samples> <outdir>   ("NULL" = absent)
// Differences by design: the correlation matrix never leaves HBM between the build and the
// sweep, adjacency comes back as a bitmap, separating sets as sparse records, only the
// retained sub-matrix is gathered to the host, and files are written with one fwrite each.
// `mps prep` / `mps block` are outside this build's scope and say so.
#include <chrono>
#include <algorithm>
#include <cstring>
#include <memory>
#include <numeric>
#include <set>

#include "../../../include/cusk_hip.h"
#include "blocking.h"
#include "host_io.h"

using namespace host;

namespace {

constexpr int ML = CUSK_ML;

[[noreturn]] void engine_die(const char *what, cusk_engine *e)
{
    std::fprintf(stderr, "mps: %s: %s\n", what, e ? cusk_last_error(e) : "no engine");
    std::exit(EXIT_FAILURE);
}

struct Bits
{
    int n = 0, words = 0;
    std::vector<uint64_t> w;
    bool get(int i, int j) const { return (w[(size_t)i * words + (j >> 6)] >> (j & 63)) & 1ull; }
};

Bits fetch_adjacency(cusk_engine *e)
{
    Bits b;
    b.n = cusk_result_n(e);
    b.words = cusk_result_words(e);
    b.w.resize((size_t)b.n * b.words);
    if (cusk_dev_download(b.w.data(), cusk_result_adj_bits_dev(e), b.w.size() * sizeof(uint64_t)) != CUSK_OK)
        engine_die("adjacency download", e);
    return b;
}

// parent_set.cpp:8-53: all traits, plus markers reached from a trait through marker nodes in
// at most max_depth hops.  Sorted ascending.
std::vector<int> subset_variables(const Bits &G, int num_var, int num_markers, int max_depth)
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

// ReducedGC / ReducedGCS of include/mps/parent_set.h
struct Reduced
{
    size_t num_var = 0, num_phen = 0, max_level = 0;
    std::vector<int> new_to_old;
    std::vector<int> G;
    std::vector<float> C;
    std::vector<float> ess;  // cuskss
    std::vector<int> S;      // cusk: num_var^2 * max_level
    size_t num_markers() const { return num_var - num_phen; }
};

void write_reduced(const Reduced &r, const std::string &base, bool with_sep)
{
    {
        std::ofstream f(base + ".mdim");
        f << r.num_var << "\t" << r.num_phen << "\t" << r.max_level << std::endl;
    }
    write_binary(base + ".ixs", r.new_to_old.data(), r.new_to_old.size());
    write_binary(base + ".adj", r.G.data(), r.G.size());
    write_binary(base + ".corr", r.C.data(), r.C.size());
    if (with_sep) write_binary(base + ".sep", r.S.data(), r.S.size());
}

std::vector<float> gather(cusk_engine *e, const float *M_dev, int n, const std::vector<int> &P)
{
    std::vector<float> out(P.size() * P.size());
    if (cusk_gather_submatrix(e, M_dev, n, P.data(), (int)P.size(), out.data()) != CUSK_OK) engine_die("gather", e);
    return out;
}

std::vector<int> gather_adj(const Bits &G, const std::vector<int> &P)
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
std::vector<int> reduce_sepsets(cusk_engine *e, const std::vector<int> &P, size_t max_level, const std::vector<int> *index_map)
{
    const size_t k = P.size();
    std::vector<int> S(k * k * max_level, -1);
    const long long cnt = cusk_result_sepsets(e, nullptr, nullptr, nullptr, nullptr, nullptr);
    if (cnt < 0) engine_die("sepsets", e);
    if (cnt == 0) return S;
    std::vector<int> x(cnt), y(cnt), rs((size_t)cnt * ML);
    if (cusk_result_sepsets(e, x.data(), y.data(), nullptr, nullptr, rs.data()) != cnt) engine_die("sepsets", e);
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

std::vector<int> compose(const std::vector<int> &P, const std::vector<int> *index_map)
{
    std::vector<int> out(P.size());
    for (size_t i = 0; i < P.size(); i++) out[i] = index_map ? (*index_map)[P[i]] : P[i];
    return out;
}

struct DevMat
{
    float *p = nullptr;
    explicit DevMat(size_t count) { p = static_cast<float *>(cusk_dev_alloc(sizeof(float) * count)); }
    DevMat(const std::vector<float> &h) : DevMat(h.size())
    {
        if (!p || cusk_dev_upload(p, h.data(), sizeof(float) * h.size()) != CUSK_OK)
        {
            std::fprintf(stderr, "mps: device upload failed\n");
            std::exit(EXIT_FAILURE);
        }
    }
    ~DevMat() { cusk_dev_free(p); }
    DevMat(const DevMat &) = delete;
    DevMat &operator=(const DevMat &) = delete;
};

// ---------------------------------------------------------------------------------------
// mps cusk   (cli.cpp:432-678)
// ---------------------------------------------------------------------------------------
const char *CUSK_USAGE = R"(
Run the skeleton search on a single block of a block diagonal genomic covariance matrix.

usage: mps cusk <.phen> <bfiles> <.blocks> <alpha> <max-level> <max-level-two> <depth> <outdir> <block-index>
)";

// wall-clock phase marks, printed as "[t] <phase>: <ms> ms" when CUSK_TIMING is set (tools/e2e_block.py)
struct PhaseTimer
{
    bool on = std::getenv("CUSK_TIMING") != nullptr;
    std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now(), last = t0;
    void mark(const char *what)
    {
        if (!on) return;
        const auto now = std::chrono::steady_clock::now();
        std::cout << "[t] " << what << ": " << std::chrono::duration<double, std::milli>(now - last).count() << " ms (at "
                  << std::chrono::duration<double, std::milli>(now - t0).count() << ")" << std::endl;
        last = now;
    }
};

int cmd_cusk(int argc, char **argv)
{
    if (argc < 11)
    {
        std::cout << CUSK_USAGE << std::endl;
        std::exit(1);
    }
    const std::string phen_path = argv[2], bfiles = argv[3], block_path = argv[4];
    const float alpha = std::stof(argv[5]);
    int max_level = std::stoi(argv[6]);
    int max_level_two = std::stoi(argv[7]);
    const int depth = std::stoi(argv[8]);
    const std::string outdir = argv[9];
    const int block_index = std::stoi(argv[10]);
    std::cout << "Got args: \n.phen: " << phen_path << "\nbfiles: " << bfiles << "\n.blocks: " << block_path
              << "\nalpha: " << alpha << "\nmax_level: " << max_level << "\nmax_level_two: " << max_level_two
              << "\ndepth: " << depth << "\noutdir: " << outdir << "\nblock-index: " << block_index << std::endl;

    PhaseTimer tm;
    std::cout << "Checking paths" << std::endl;
    for (const char *sfx : {".bed", ".dim", ".means", ".stds", ".bim"}) check_path(bfiles + sfx);
    if (!bed_has_valid_magic(bfiles + ".bed")) die("unexpected magic number in bed file.");
    check_path(phen_path);
    check_path(block_path);
    check_path(outdir);

    Phen phen = load_phen(phen_path);
    tm.mark("load phen");
    const BedDims dims = read_dims(bfiles + ".dim");
    if (phen.num_samples != dims.num_samples) die("different num samples in phen and dims");
    const BimInfo bim = read_bim(bfiles + ".bim");
    const size_t N = dims.num_samples, p = phen.num_phen;
    std::cout << "Found " << p << " phenotypes" << std::endl;
    std::cout << "Loading blocks" << std::endl;
    const std::vector<Block> blocks = read_blocks(block_path);
    std::cout << "Found " << blocks.size() << " blocks" << std::endl;
    for (const Block &b : blocks)
        if (b.first >= bim.markers_on(b.chr) || b.last >= bim.markers_on(b.chr))
            die("block out of bounds with first_ix: " + std::to_string(b.first) + " last_ix: " + std::to_string(b.last));
    if (block_index < 0 || (size_t)block_index >= blocks.size()) die("block index out of range");

    float Th[ML + 1];
    cusk_threshold_array((int)N, alpha, Th);
    std::cout << "Number of levels: " << max_level << std::endl;
    std::cout << "Setting level thr for cuPC: " << std::endl;
    for (int i = 0; i <= std::min(max_level, ML); ++i) std::cout << "\t Level: " << i << " thr: " << Th[i] << std::endl;

    const Block block = blocks[block_index];
    const size_t m = block.size();
    std::cout << "\nProcessing block " << block_index + 1 << " / " << blocks.size() << std::endl;
    std::cout << "Block size: " << m << std::endl;
    std::cout << "Loading bed data" << std::endl;
    const std::vector<unsigned char> bed = read_bed_block(bfiles + ".bed", block, dims, bim);
    const size_t g0 = bim.start_of(block.chr) + block.first, g1 = bim.start_of(block.chr) + block.last;
    const std::vector<float> means = read_floats_line_range(bfiles + ".means", g0, g1);
    const std::vector<float> stds = read_floats_line_range(bfiles + ".stds", g0, g1);
    if (means.size() != m || stds.size() != m) die("block size and number of means or stds differ");
    tm.mark("bim, blocks, bed block, means, stds");

    cusk_engine *e = nullptr;
    if (cusk_engine_create(&e, 0, nullptr) != CUSK_OK) engine_die("engine create (is a HIP device visible?)", nullptr);
    const size_t n = m + p;
    DevMat C(n * n);
    if (!C.p) engine_die("device allocation", e);
    tm.mark("engine create + device allocation");

    std::cout << "Checking for significant marker - phen correlations" << std::endl;
    std::cout << "Computing all correlations" << std::endl;
    std::vector<float> mxp(m * p);
    if (cusk_corr_build(e, bed.data(), phen.data.data(), m, N, p, means.data(), stds.data(), C.p, mxp.data()) != CUSK_OK)
        engine_die("correlation build", e);
    tm.mark("correlation build (H2D + kernels + mxp D2H)");
    // cli.cpp:561-576: blocks without any marginally significant marker-trait correlation are skipped
    int num_sig = 0;
    for (float c : mxp) num_sig += (std::fabs(0.5 * (std::log(std::fabs((1 + c))) - std::log(std::fabs(1 - c)))) >= Th[0]);
    if (num_sig > 0)
        std::cout << "Found " << num_sig << " marker - phen correlations. Proceeding." << std::endl;
    else
    {
        std::cout << "No significant correlations found. Skipping block." << std::endl;
        cusk_engine_destroy(e);
        return 0;
    }
    if (std::getenv("CUSK_WRITE_FULL_CORRMATS"))
    {  // cli.cpp:27,651-658 (compile-time switch in the reference)
        std::vector<float> full(n * n);
        cusk_dev_download(full.data(), C.p, sizeof(float) * n * n);
        write_binary(make_path(outdir, block.file_stem(), ".all_corrs"), full.data(), full.size());
    }

    std::cout << "Running cuPC" << std::endl;
    cusk_engine_set_option(e, "assume_symmetric", 1);  // cusk_corr_build mirrors every element
    cusk_stats st;
    if (cusk_run_skeleton(e, C.p, (int)n, Th, max_level, &st) != CUSK_OK) engine_die("Skeleton", e);
    tm.mark("skeleton stage one");
    for (int l = 0; l < st.levels_run; l++)
        std::cout << "level " << l << ": max degree " << st.max_degree[l] << ", " << st.tests[l] << " tests, "
                  << st.level_ms[l] * 1e-3 << " s" << std::endl;
    Bits G = fetch_adjacency(e);
    std::vector<int> P = subset_variables(G, (int)n, (int)m, depth);
    Reduced gcs;
    gcs.num_var = P.size();
    gcs.num_phen = p;
    gcs.max_level = (size_t)max_level;
    gcs.new_to_old = P;
    gcs.C = gather(e, C.p, (int)n, P);
    tm.mark("adjacency fetch + prune + sub-matrix gather");
    // (the stage-one separating sets of cli.cpp:673 are never read again: stage two recomputes them)

    std::cout << "Starting second cusk stage" << std::endl;
    {  // cli.cpp:62-87: Skeleton again on the reduced set, starting from the complete graph
        const int k = (int)gcs.num_var;
        DevMat C2(gcs.C);
        cusk_engine_set_option(e, "assume_symmetric", 0);
        if (cusk_run_skeleton(e, C2.p, k, Th, max_level_two, &st) != CUSK_OK) engine_die("Skeleton (stage two)", e);
        tm.mark("skeleton stage two");
        if (tm.on)
            for (int l = 0; l < st.levels_run; l++)
                std::cout << "[t] stage two level " << l << ": max degree " << st.max_degree[l] << ", " << st.edges[l]
                          << " edges, " << st.tests[l] << " tests, " << st.subsets[l] << " sets, " << st.rechecks[l]
                          << " rechecks, sweep " << st.kernel_ms[l] << " ms, level " << st.level_ms[l] << " ms" << std::endl;
        Bits G2 = fetch_adjacency(e);
        std::vector<int> P2 = subset_variables(G2, k, (int)gcs.num_markers(), depth);
        Reduced out;
        out.num_var = P2.size();
        out.num_phen = p;
        out.max_level = ML;
        out.new_to_old = compose(P2, &gcs.new_to_old);
        out.G = gather_adj(G2, P2);
        out.C = gather(e, C2.p, k, P2);
        out.S = reduce_sepsets(e, P2, ML, &gcs.new_to_old);
        std::cout << "Retained " << out.num_markers() << " markers" << std::endl;
        tm.mark("stage-two reduction");
        write_reduced(out, make_path(outdir, block.file_stem(), ""), true);
        tm.mark("write outputs");
    }
    cusk_engine_destroy(e);
    tm.mark("engine destroy");
    return 0;
}

// ---------------------------------------------------------------------------------------
// mps cuskss   (mps.cpp:31-101, cli.cpp:29-60, :89-173, :194-346)
// ---------------------------------------------------------------------------------------
struct GC
{
    size_t num_var = 0, num_phen = 0;
    std::vector<int> new_to_old;
    std::vector<int> G;      // empty = complete graph
    std::vector<float> C;    // num_var^2
    std::vector<float> ess;  // num_var^2 when heterogeneous, else empty
    size_t num_markers() const { return num_var - num_phen; }
};

// run_cusk, cli.cpp:29-60: hetcor_skeleton, prune to depth, extract the retained sub-matrices
GC run_cusk(cusk_engine *e, const GC &gc, float th, float ess_uniform, bool het, int depth, int max_level,
            const std::vector<int> &time_index_traits)
{
    const int n = (int)gc.num_var;
    std::vector<int> ti(n, 0);
    for (size_t i = gc.num_markers(), r = 0; i < gc.num_var; i++, r++) ti[i] = time_index_traits[r];
    DevMat C(gc.C);
    std::unique_ptr<DevMat> Nd;
    if (het) Nd.reset(new DevMat(gc.ess));
    int *Gd = nullptr;
    if (!gc.G.empty())
    {
        Gd = static_cast<int *>(cusk_dev_alloc(sizeof(int) * gc.G.size()));
        if (!Gd || cusk_dev_upload(Gd, gc.G.data(), sizeof(int) * gc.G.size()) != CUSK_OK) engine_die("upload G", e);
    }
    cusk_stats st;
    if (cusk_run_hetcor(e, C.p, het ? Nd->p : nullptr, ess_uniform, Gd, n, th, max_level, ti.data(), &st) != CUSK_OK)
        engine_die("hetcor_skeleton", e);
    cusk_dev_free(Gd);
    Bits G = fetch_adjacency(e);
    std::vector<int> P = subset_variables(G, n, (int)gc.num_markers(), depth);
    GC out;
    out.num_var = P.size();
    out.num_phen = gc.num_phen;
    out.new_to_old = compose(P, &gc.new_to_old);
    out.G = gather_adj(G, P);
    out.C = gather(e, C.p, n, P);
    if (het) out.ess = gather(e, Nd->p, n, P);
    return out;
}

void write_gc(const GC &gc, const std::string &base)
{
    Reduced r;
    r.num_var = gc.num_var;
    r.num_phen = gc.num_phen;
    r.max_level = ML;  // cli.cpp:58 passes ML
    r.new_to_old = gc.new_to_old;
    r.G = gc.G;
    r.C = gc.C;
    write_reduced(r, base, false);
}

int cmd_cuskss(int argc, char **argv)
{
    if (argc < 17) die("usage: mps cuskss <mxm> <mxp> <mxp_se> <pxp> <pxp_se> <time_index> <block_index> <blockfile> "
                       "<marker_indices> <alpha> <max_level_one> <max_level_two> <depth> <num_samples> <outdir>");
    const std::string mxm_path = argv[2], mxp_path = argv[3], mxp_se_path = argv[4], pxp_path = argv[5],
                      pxp_se_path = argv[6], time_index_path = argv[7];
    const int block_index = std::stoi(argv[8]);
    const std::string blockfile_path = argv[9], marker_index_path = argv[10];
    const float alpha = std::stof(argv[11]);
    const int max_level_one = std::stoi(argv[12]), max_level_two = std::stoi(argv[13]), depth = std::stoi(argv[14]);
    const float num_samples = (float)std::stoi(argv[15]);
    const std::string outdir = argv[16];
    const bool merged = marker_index_path != "NULL", hetcor = mxp_se_path != "NULL", trait_only = mxm_path == "NULL",
               two_stage = max_level_two > 0, time_indexed = time_index_path != "NULL";
    check_path(pxp_path);
    check_path(outdir);
    if (merged)
        check_path(marker_index_path);
    else
        check_path(blockfile_path);
    if (hetcor || merged || !trait_only)
    {
        check_path(mxm_path);
        check_path(mxp_path);
    }
    if (hetcor)
    {
        check_path(mxp_se_path);
        check_path(pxp_se_path);
    }
    if (time_indexed) check_path(time_index_path);

    std::cout << "Loading input files" << std::endl;
    Block block;
    std::vector<size_t> rows;
    if (merged)
    {
        std::cout << "Loading marker indices" << std::endl;
        for (int v : read_binary<int>(marker_index_path)) rows.push_back((size_t)v);
    }
    else
    {
        std::cout << "Loading block file" << std::endl;
        const std::vector<Block> blocks = read_blocks(blockfile_path);
        if (block_index < 0 || (size_t)block_index >= blocks.size()) die("block index out of range");
        block = blocks[block_index];
        for (size_t r = block.first_global(); r <= block.last_global(); r++) rows.push_back(r);
    }
    std::cout << "Loading pxp" << std::endl;
    const Pxp pxp = load_pxp(pxp_path, hetcor ? pxp_se_path : std::string(), num_samples);
    const size_t p = pxp.num_phen;
    std::vector<int> time_index_traits(p, 1);
    if (time_indexed)
    {
        std::cout << "Loading time_indices" << std::endl;
        time_index_traits = read_ints_lines(time_index_path);
        if (time_index_traits.size() < p) die("time index file has fewer lines than traits");
    }
    const float th = cusk_hetcor_threshold(alpha);
    cusk_engine *e = nullptr;
    if (cusk_engine_create(&e, 0, nullptr) != CUSK_OK) engine_die("engine create (is a HIP device visible?)", nullptr);

    GC gc;
    gc.num_phen = p;
    if (trait_only)
    {  // cli.cpp:225-256
        gc.num_var = p;
        gc.C = pxp.corr;
        if (hetcor) gc.ess = pxp.ess;
        gc.new_to_old.resize(p);
        std::iota(gc.new_to_old.begin(), gc.new_to_old.end(), 0);
        std::cout << "Starting first cusk stage" << std::endl;
        gc = run_cusk(e, gc, th, num_samples, hetcor, depth, max_level_one, time_index_traits);
        write_gc(gc, make_path(outdir, "trait_only", ""));
        std::cout << "Retained " << gc.num_markers() << " markers" << std::endl;
        cusk_engine_destroy(e);
        return 0;
    }
    std::cout << "Loading mxm" << std::endl;
    const size_t m = mxm_num_markers(mxm_path);
    std::cout << "Loading mxp summary stats" << std::endl;
    const Mxp mxp = load_mxp(mxp_path, hetcor ? mxp_se_path : std::string(), rows);
    if (pxp.num_phen != mxp.num_phen) die("Numbers of traits seem to differ between pxp and mxp");
    if (m != mxp.num_markers)
        die("Numbers of markers seem to differ between mxm and mxp\nmxp: " + std::to_string(mxp.num_markers) + " x " +
            std::to_string(mxp.num_phen) + "\nmxm: " + std::to_string(m) + " x " + std::to_string(m));
    std::cout << "Merging correlations into single matrix" << std::endl;
    // make_square_cuskss_inputs, cli.cpp:89-173: markers first, then traits
    const size_t n = m + p;
    gc.num_var = n;
    gc.C.assign(n * n, 1.0f);
    load_mxm_into(mxm_path, m, gc.C.data(), n);
    if (hetcor) gc.ess.assign(n * n, num_samples);
    for (size_t i = 0; i < m; i++)
        for (size_t k = 0; k < p; k++)
        {
            gc.C[i * n + m + k] = gc.C[(m + k) * n + i] = mxp.corr[i * p + k];
            if (hetcor) gc.ess[i * n + m + k] = gc.ess[(m + k) * n + i] = mxp.ess[i * p + k];
        }
    for (size_t a = 0; a < p; a++)
        for (size_t b = 0; b < p; b++)
        {
            gc.C[(m + a) * n + m + b] = pxp.corr[a * p + b];
            if (hetcor) gc.ess[(m + a) * n + m + b] = pxp.ess[a * p + b];
        }
    gc.new_to_old.resize(n);
    std::iota(gc.new_to_old.begin(), gc.new_to_old.end(), 0);
    std::cout << "Starting first cusk stage" << std::endl;
    gc = run_cusk(e, gc, th, num_samples, hetcor, depth, max_level_one, time_index_traits);
    if (two_stage)
    {
        std::cout << "Starting second cusk stage" << std::endl;
        gc = run_cusk(e, gc, th, num_samples, hetcor, depth, max_level_two, time_index_traits);
    }
    std::cout << "Retained " << gc.num_markers() << " markers" << std::endl;
    write_gc(gc, make_path(outdir, merged ? "cuskss_merged" : block.file_stem(), ""));
    cusk_engine_destroy(e);
    return 0;
}

const char *MPS_USAGE = R"(
usage: mps <command> [<args>]

commands:
    cusk                    Run the skeleton search on a single block of block diagonal genomic covariance matrix
    cuskss                  Run the skeleton search on a block of markers and traits with pre-computed correlations.
    block                   Tile the marker x marker correlation matrix of every chromosome into LD blocks
    prep                    Prepare input (PLINK) .bed file for cusk: .dim, .means, .stds, .modes
)";

}  // namespace

// ---------------------------------------------------------------------------------------
// mps block   (cli.cpp:348-411)
// ---------------------------------------------------------------------------------------
const char *BLOCK_USAGE = R"(
Tile marker x marker correlation matrix

usage: mps block <bfiles> <max-block-size> <device-mem-gb> <corr-width>

arguments:
    bfiles          stem of .bed, .bim, .fam fileset
    max-block-size  maximum number of markers per block
    device-mem-gb   maximum memory available on gpu in GB
    corr-width      max distance at which to compute correlations
)";

int cmd_block(int argc, char **argv)
{
    if (argc < 6)
    {
        std::cout << BLOCK_USAGE << std::endl;
        std::exit(1);
    }
    const std::string bfiles = argv[2];
    const int max_block_size = std::stoi(argv[3]);
    const float device_mem_gb = std::stof(argv[4]);
    const size_t corr_width = (size_t)std::stoi(argv[5]);
    PhaseTimer tm;

    std::cout << "Checking paths" << std::endl;
    for (const char *sfx : {".bed", ".bim", ".fam"}) check_path(bfiles + sfx);
    if (!bed_has_valid_magic(bfiles + ".bed")) die("unexpected magic number in bed file.");
    const BimInfo bim = read_bim(bfiles + ".bim");
    const size_t N = count_lines(bfiles + ".fam");  // BedDims(BfilesBase), io.h:41-45
    const size_t bytes_per_marker = (N + 3) / 4;
    const std::string out_path = bfiles + "_m" + std::to_string(max_block_size) + ".blocks";  // bfiles_base.h:35

    cusk_engine *e = nullptr;
    if (cusk_engine_create(&e, 0, nullptr) != CUSK_OK) engine_die("engine create (is a HIP device visible?)", nullptr);
    for (const std::string &cid : bim.chr_ids)
    {
        const size_t m = bim.markers_on(cid);
        const size_t mem_host_gb = (size_t)(((double)(m * bytes_per_marker) + (double)corr_width * (double)m * 4.0) * 1e-9);
        std::cout << "[Chr " << cid << "]: At least " << mem_host_gb << " GB in host memory required." << std::endl;
        std::cout << "[Chr " << cid << "]: Loading bed data for " << m << " markers." << std::endl;
        Block whole;
        whole.chr = cid;
        whole.first = 0;
        whole.last = m - 1;
        BedDims dims;
        dims.num_samples = N;
        dims.num_markers = bim.num_lines;
        const std::vector<unsigned char> bed = read_bed_block(bfiles + ".bed", whole, dims, bim);
        tm.mark("read chromosome");

        std::cout << "[Chr " << cid << "]: Computing correlations." << std::endl;
        {  // corr_host.cu:73-91: the reference sizes a marker batch from the device memory it is told about
            const double mem_bytes = (double)device_mem_gb * 1e9;
            size_t batch = (size_t)std::floor(mem_bytes / (double)(bytes_per_marker + corr_width * 4));
            if (batch > m) batch = m;
            if (batch < corr_width)
            {
                std::printf("Maximal batch size (%zu) < corr width (%zu). Decrease distance threshold or increase device memory. \n",
                            batch, corr_width);
                std::exit(1);
            }
        }
        std::vector<float> row_sums(m);
        if (cusk_corr_banded(e, bed.data(), m, N, corr_width, row_sums.data(), nullptr) != CUSK_OK)
            engine_die("banded correlations", e);
        tm.mark("banded correlations + row sums (device)");
        std::cout << "[Chr " << cid << "]: Computing row sums." << std::endl;
        std::cout << "[Chr " << cid << "]: Making blocks." << std::endl;
        const std::vector<ChrBlock> blocks =
            blocks_of_chromosome(row_sums, max_block_size, [&](const std::vector<float> &v, const std::vector<double> &w) {
                std::vector<double> out(v.size());
                if (cusk_hanning_smooth(e, v.data(), v.size(), w.data(), (int)w.size(), out.data()) != CUSK_OK)
                    engine_die("Hanning smoothing", e);
                return out;
            });
        tm.mark("smoothing (device) + minima + bisection");
        std::cout << "[Chr " << cid << "]: Partitioned into " << blocks.size() << " blocks." << std::endl;
        std::cout << "[Chr " << cid << "]: Writing blocks to output file." << std::endl;
        std::ofstream fout(out_path, std::ios::out | std::ios::app);  // io.cpp:266-277: appends
        for (const ChrBlock &b : blocks) fout << cid << "\t" << b.first << "\t" << b.last << std::endl;
    }
    cusk_engine_destroy(e);
    std::cout << "Done." << std::endl;
    return 0;
}

// ---------------------------------------------------------------------------------------
// mps prep   (cli.cpp:680-708, prep.cpp:15-76 compute_bed_col_stats_no_impute, :159-203 prep_bed_no_impute)
// ---------------------------------------------------------------------------------------
// Per marker over its non-missing genotypes: mean = sum / (float)count, population standard deviation with the
// squared deviations accumulated in single precision in sample order, most frequent genotype (ties to the lower
// one).  Host code like the reference's (one streaming pass over the .bed; nothing here is worth a device launch);
// the arithmetic is kept operation for operation because cusk's Pearson correlations divide by these numbers.
const char *PREP_USAGE = R"(
Prepare input (PLINK) .bed file for cusk

usage: mps prep <.bfiles>

arguments:
    .bfiles filestem of .bed, .bim, .fam fileset
)";

int cmd_prep(int argc, char **argv)
{
    if (argc != 3 || std::string(argv[2]) == "--help" || std::string(argv[2]) == "-h")
    {
        std::cout << PREP_USAGE << std::endl;
        std::exit(1);
    }
    const std::string bfiles = argv[2];
    for (const char *sfx : {".bed", ".bim", ".fam"}) check_path(bfiles + sfx);
    if (!bed_has_valid_magic(bfiles + ".bed"))
    {
        std::cout << "Invalid prefix bytes in bed" << std::endl;
        std::exit(1);
    }
    const size_t N = count_lines(bfiles + ".fam");
    const size_t M = count_lines(bfiles + ".bim");
    std::cout << "Writing dim file." << std::endl;
    {
        std::ofstream fout(bfiles + ".dim");
        fout << N << "\t" << M << std::endl;
    }
    const size_t bytes_per_marker = (N + 3) / 4;
    std::vector<unsigned char> col(bytes_per_marker);
    std::vector<float> means, stds;
    std::vector<int> modes;
    std::ifstream bed(bfiles + ".bed", std::ios::binary);
    bed.seekg(3);
    std::cout << "Computing means, stds, modes." << std::endl;
    // PLINK codes, low bits first (bed_lut.h): 00 -> 2, 01 -> missing, 10 -> 1, 11 -> 0
    static const int kValue[4] = {2, 0, 1, 0};
    size_t lc = 0;
    while (bed.read(reinterpret_cast<char *>(col.data()), (std::streamsize)bytes_per_marker))
    {
        if (lc % 100000 == 0) std::cout << "Processing marker " << lc + 1 << " / " << M << std::endl;
        int counts[3] = {0, 0, 0};
        size_t sum = 0, missing = 0;
        for (size_t i = 0; i < N; i++)
        {
            const int code = (col[i >> 2] >> (2 * (i & 3))) & 3;
            if (code == 1)
                missing++;
            else
            {
                counts[kValue[code]]++;
                sum += (size_t)kValue[code];
            }
        }
        int mode = 0;
        for (int g = 1; g < 3; g++)
            if (counts[g] > counts[mode]) mode = g;
        const float mean = sum / (float)(N - missing);
        float ss = 0.0f;
        for (size_t i = 0; i < N; i++)
        {
            const int code = (col[i >> 2] >> (2 * (i & 3))) & 3;
            if (code != 1) ss += ((float)kValue[code] - mean) * ((float)kValue[code] - mean);
        }
        means.push_back(mean);
        stds.push_back(std::sqrt(ss / (float)(N - missing)));
        modes.push_back(mode);
        lc++;
    }
    std::cout << "Writing stats to files." << std::endl;
    {
        std::ofstream f(bfiles + ".means");
        for (float v : means) f << v << std::endl;
    }
    {
        std::ofstream f(bfiles + ".stds");
        for (float v : stds) f << v << std::endl;
    }
    {
        std::ofstream f(bfiles + ".modes");
        for (int v : modes) f << v << std::endl;
    }
    std::cout << "Done." << std::endl;
    return 0;
}

int main(int argc, char **argv)
{
    if (argc == 1)
    {
        std::cout << MPS_USAGE << std::endl;
        return EXIT_SUCCESS;
    }
    const std::string cmd = argv[1];
    if (cmd == "cusk") return cmd_cusk(argc, argv);
    if (cmd == "cuskss") return cmd_cuskss(argc, argv);
    if (cmd == "block") return cmd_block(argc, argv);
    if (cmd == "prep") return cmd_prep(argc, argv);
    std::cout << MPS_USAGE << std::endl;
    return EXIT_SUCCESS;
}
