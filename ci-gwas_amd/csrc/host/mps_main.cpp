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
#include "block_pipeline.h"
#include "blocking.h"
#include "host_io.h"

using namespace host;

namespace {

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
    CuskInputs in;
    in.phen_path = argv[2];
    in.bfiles = argv[3];
    in.block_path = argv[4];
    in.alpha = std::stof(argv[5]);
    in.max_level = std::stoi(argv[6]);
    in.max_level_two = std::stoi(argv[7]);
    in.depth = std::stoi(argv[8]);
    const std::string outdir = argv[9];
    const int block_index = std::stoi(argv[10]);
    std::cout << "Got args: \n.phen: " << in.phen_path << "\nbfiles: " << in.bfiles << "\n.blocks: " << in.block_path
              << "\nalpha: " << in.alpha << "\nmax_level: " << in.max_level << "\nmax_level_two: " << in.max_level_two
              << "\ndepth: " << in.depth << "\noutdir: " << outdir << "\nblock-index: " << block_index << std::endl;

    PhaseTimer tm;
    check_path(outdir);
    in.load(&std::cout);  // cli.cpp:458-497
    tm.mark("load phen, dim, bim, blocks; map bed");
    if (block_index < 0 || (size_t)block_index >= in.blocks.size()) die("block index out of range");
    std::cout << "Number of levels: " << in.max_level << std::endl;
    std::cout << "Setting level thr for cuPC: " << std::endl;
    for (int i = 0; i <= std::min(in.max_level, ML); ++i) std::cout << "\t Level: " << i << " thr: " << in.Th[i] << std::endl;
    if (std::getenv("CUSK_WRITE_FULL_CORRMATS")) in.full_corrmats_dir = outdir;

    cusk_engine *e = nullptr;
    if (cusk_engine_create(&e, 0, nullptr) != CUSK_OK) engine_die("engine create (is a HIP device visible?)", nullptr);
    tm.mark("engine create");
    // the one block of this invocation through the pipeline the block driver runs for every block of a chromosome
    BlockScratch scratch;
    Reduced out;
    std::string stem;
    BlockStats bs;
    const bool kept = run_cusk_block(e, in, block_index, scratch, out, stem, bs, &std::cout);
    if (tm.on)
    {
        std::cout << "[t] inputs (bed slice, means, stds): " << bs.ms_inputs << " ms\n[t] correlation build (H2D + kernels + mxp D2H): "
                  << bs.ms_corr << " ms" << std::endl;
        if (kept)
        {
            std::cout << "[t] skeleton stage one: " << bs.ms_stage1 << " ms\n[t] adjacency fetch + prune + sub-matrix gather: "
                      << bs.ms_prune << " ms\n[t] skeleton stage two: " << bs.ms_stage2 << " ms\n[t] stage-two reduction: "
                      << bs.ms_reduce << " ms" << std::endl;
            const cusk_stats &st = bs.stage[1];
            for (int l = 0; l < st.levels_run; l++)
                std::cout << "[t] stage two level " << l << ": max degree " << st.max_degree[l] << ", " << st.edges[l]
                          << " edges, " << st.tests[l] << " tests, " << st.subsets[l] << " sets, " << st.rechecks[l]
                          << " rechecks, sweep " << st.kernel_ms[l] << " ms, level " << st.level_ms[l] << " ms" << std::endl;
            std::cout << "[t] filter contradictions (option validate): stage one " << bs.stage[0].violations << ", stage two "
                      << st.violations << "; exact fallbacks " << bs.stage[0].exact_fallbacks << " / " << st.exact_fallbacks << std::endl;
        }
        tm.mark("block pipeline");
    }
    if (kept)
    {
        write_reduced(out, make_path(outdir, stem, ""), true);
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
    try
    {
        if (cmd == "cusk") return cmd_cusk(argc, argv);
        if (cmd == "cuskss") return cmd_cuskss(argc, argv);
        if (cmd == "block") return cmd_block(argc, argv);
        if (cmd == "prep") return cmd_prep(argc, argv);
    }
    catch (const Fatal &f)
    {  // bad input: message + status 1, as the reference's loaders and argument checks do (cli.cpp:185-192)
        std::cerr << f.what() << std::endl;
        return 1;
    }
    catch (const EngineError &f)
    {  // device failure: gpuerrors.h:6-15
        std::fprintf(stderr, "mps: %s\n", f.what());
        return EXIT_FAILURE;
    }
    std::cout << MPS_USAGE << std::endl;
    return EXIT_SUCCESS;
}
