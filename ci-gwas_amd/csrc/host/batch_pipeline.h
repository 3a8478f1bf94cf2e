// batch_pipeline.h -- MANY LD blocks through the `cusk` pipeline in one set of device runs (include/cusk_hip.h section 3:
// cusk_blockset_run_batch).
//
// The reference runs one block per `mps cusk` process (/root/reference/cusk/src/cli.cpp:507-512, README.md:62); a
// 500-SNP block is a chain of launch-latency-bound kernels that leaves an MI355X idle.  Here the blocks a GPU owns go
// through cli.cpp:521-677 TOGETHER: their correlation matrices are built by one set of launches onto the diagonal of one
// allocation (cusk_corr_build_batch_*), stage one sweeps them in ONE level loop (cusk_run_skeleton_batch), the pruned
// sub-matrices are gathered by one launch onto the diagonal of a second allocation, stage two sweeps those in one level
// loop, and one read-out brings adjacency, separating sets and correlations back.  Per block the result is the one of
// run_cusk_block (block_pipeline.h) -- the files are byte-identical (tests/test_gpu_batch.py) -- because the blocks never
// interact: a row only meets columns of its own block.
#pragma once
#include <thread>
#include <cstdio>
#include <cstdlib>

#include "block_pipeline.h"

namespace host {

struct BatchScratch
{
    DevMat C, C2;
};

struct BatchStats
{
    int blocks = 0, skipped = 0;
    long long markers = 0, retained = 0;
    long long vars_stage1 = 0, vars_stage2 = 0;  // padded variable counts of the two batch allocations
    long long tests[2] = {0, 0}, canonical[2] = {0, 0};
    double ms_corr = 0, ms_stage1 = 0, ms_prune = 0, ms_stage2 = 0, ms_reduce = 0;
    cusk_stats stage[2];
};

struct BatchBlockOut
{
    int block_index = -1;
    bool skipped = false;
    int num_sig = 0;
    std::string stem;
    Reduced r;
};

inline int pad64(size_t v) { return (int)((v + 63) / 64 * 64); }

// rows lo..hi-1 of one block out of the packed bitmap of cusk_result_adj_bits_blocks
inline Bits block_bits(const std::vector<uint64_t> &packed, size_t &pos, int k)
{
    Bits b;
    b.n = k;
    b.words = (k + 63) / 64;
    b.w.assign(packed.begin() + (long)pos, packed.begin() + (long)(pos + (size_t)k * b.words));
    pos += (size_t)k * b.words;
    return b;
}

// cli.cpp:521-677 for every block of `blocks` on the engine's device.  outs[i] belongs to blocks[i].
inline void run_cusk_batch(cusk_engine *e, const CuskInputs &in, const StagedInputs &staged, const std::vector<int> &blocks,
                           BatchScratch &scr, std::vector<BatchBlockOut> &outs, BatchStats &bs)
{
    using clk = std::chrono::steady_clock;
    auto ms_since = [](clk::time_point &t) {
        const auto now = clk::now();
        const double v = std::chrono::duration<double, std::milli>(now - t).count();
        t = now;
        return v;
    };
    bs = BatchStats();
    // CUSK_BATCH_PROF=1: wall-clock marks of this function on stderr (microseconds since entry)
    static const bool prof = std::getenv("CUSK_BATCH_PROF") != nullptr;
    const auto p0 = clk::now();
    std::string plog;
    auto mark = [&](const char *what) {
        if (!prof) return;
        char buf[64];
        std::snprintf(buf, sizeof(buf), " %s=%.0f", what, std::chrono::duration<double, std::micro>(clk::now() - p0).count());
        plog += buf;
    };
    const int B = (int)blocks.size();
    outs.assign((size_t)B, BatchBlockOut());
    if (B == 0) return;
    if (cusk_engine_bind_thread(e) != CUSK_OK) engine_die("bind thread", e);
    const size_t N = in.dims.num_samples, p = in.phen.num_phen, bpc = in.dims.bytes_per_col();
    auto t = clk::now();

    // ---- layout of stage one: block b = variables [base[b], base[b] + m_b + p), bases multiples of 64 ----
    std::vector<long long> first((size_t)B);
    std::vector<int> m((size_t)B), base((size_t)B);
    size_t n1 = 0, msum = 0;
    for (int b = 0; b < B; b++)
    {
        const int bi = blocks[(size_t)b];
        if (bi < 0 || (size_t)bi >= in.blocks.size()) die("block index out of range");
        const Block &blk = in.blocks[(size_t)bi];
        outs[(size_t)b].block_index = bi;
        outs[(size_t)b].stem = blk.file_stem();
        const size_t g0 = in.first_marker(blk);
        if (3 + (g0 + blk.size()) * bpc > in.bed.size) die("bed file is shorter than .dim / .bim say");
        first[(size_t)b] = (long long)g0;
        m[(size_t)b] = (int)blk.size();
        base[(size_t)b] = (int)n1;
        n1 += (size_t)pad64(blk.size() + p);
        msum += blk.size();
    }
    if (n1 > (size_t)0x7fffffff) die("batch too large");
    bs.blocks = B;
    bs.markers = (long long)msum;
    bs.vars_stage1 = (long long)n1;
    scr.C.reserve(n1 * n1);

    // ---- correlations: marker x trait first (prefilter, cli.cpp:550-576), the rest for the blocks that pass ----
    // (the marker x marker part is enqueued for every block at once and runs while the host does the prefilter)
    std::vector<float> mxp(msum * p);
    if (cusk_corr_build_batch(e, staged.bed, staged.phen, staged.means, staged.stds, N, p, B, first.data(), m.data(), base.data(), (int)n1,
                              scr.C.p, mxp.data()) != CUSK_OK)
        engine_die("correlation build", e);
    mark("mxp");
    std::vector<unsigned char> keep((size_t)B, 0);
    std::vector<int> kept;
    {
        size_t o = 0;
        for (int b = 0; b < B; b++)
        {
            const size_t cnt = (size_t)m[(size_t)b] * p;
            const int num_sig = count_significant(mxp.data() + o, cnt, in.Th[0]);
            o += cnt;
            outs[(size_t)b].num_sig = num_sig;
            outs[(size_t)b].skipped = (num_sig == 0);
            keep[(size_t)b] = num_sig > 0;
            if (num_sig > 0)
                kept.push_back(b);
            else
                bs.skipped++;
        }
    }
    mark("prefilter");
    if (kept.empty())
    {
        bs.ms_corr = ms_since(t);
        return;
    }
    mark("mxm_enq");
    bs.ms_corr = ms_since(t);

    // ---- stage one: one level loop for every kept block ----
    const int K = (int)kept.size();
    std::vector<int> lo1((size_t)K), hi1((size_t)K);
    for (int k = 0; k < K; k++)
    {
        lo1[(size_t)k] = base[(size_t)kept[(size_t)k]];
        hi1[(size_t)k] = lo1[(size_t)k] + m[(size_t)kept[(size_t)k]] + (int)p;
    }
    if (cusk_run_skeleton_batch(e, scr.C.p, (int)n1, K, lo1.data(), hi1.data(), in.Th, in.max_level, &bs.stage[0]) != CUSK_OK)
        engine_die("Skeleton (batch)", e);
    mark("stage1");
    for (int l = 0; l < bs.stage[0].levels_run; l++)
    {
        bs.tests[0] += bs.stage[0].tests[l];
        bs.canonical[0] += bs.stage[0].canonical_tests[l];
    }
    bs.ms_stage1 = ms_since(t);

    // ---- prune (parent_set.cpp:8-53 per block) and the stage-two matrices (cli.cpp:62-87), gathered device to device ----
    std::vector<uint64_t> packed;
    const bool traits_only = (in.depth == 1 && p > 0);  // depth 1 looks at the trait rows only
    {
        size_t words = 0;
        for (int k = 0; k < K; k++)
            words += (size_t)(traits_only ? (int)p : hi1[(size_t)k] - lo1[(size_t)k]) * (size_t)((hi1[(size_t)k] - lo1[(size_t)k] + 63) / 64);
        packed.resize(words);
        if ((traits_only ? cusk_result_adj_bits_blocks_tail(e, (int)p, packed.data()) : cusk_result_adj_bits_blocks(e, packed.data())) != CUSK_OK)
            engine_die("adjacency (batch)", e);
    }
    mark("bits1");
    std::vector<std::vector<int>> P1((size_t)K);
    std::vector<int> lo2((size_t)K), hi2((size_t)K);
    size_t n2 = 0, rows2 = 0;
    {
        size_t pos = 0;
        for (int k = 0; k < K; k++)
        {
            const int nb = hi1[(size_t)k] - lo1[(size_t)k];
            if (traits_only)
            {
                Bits T;
                T.n = (int)p;
                T.words = (nb + 63) / 64;
                T.w.assign(packed.begin() + (long)pos, packed.begin() + (long)(pos + (size_t)T.n * T.words));
                pos += (size_t)T.n * T.words;
                P1[(size_t)k] = subset_variables_depth1(T, nb, nb - (int)p);
            }
            else
            {
                const Bits G = block_bits(packed, pos, nb);
                P1[(size_t)k] = subset_variables(G, nb, nb - (int)p, in.depth);
            }
            lo2[(size_t)k] = (int)n2;
            hi2[(size_t)k] = (int)n2 + (int)P1[(size_t)k].size();
            n2 += (size_t)pad64(P1[(size_t)k].size());
            rows2 += P1[(size_t)k].size();
        }
    }
    mark("bfs1");
    bs.vars_stage2 = (long long)n2;
    scr.C2.reserve(n2 * n2);
    {
        std::vector<int> idx(rows2), row_src(rows2), row_k(rows2);
        std::vector<long long> row_first(rows2), row_out(rows2);
        size_t r = 0;
        for (int k = 0; k < K; k++)
        {
            const size_t f0 = r;
            const int kk = (int)P1[(size_t)k].size();
            for (int i = 0; i < kk; i++, r++)
            {
                idx[r] = lo1[(size_t)k] + P1[(size_t)k][(size_t)i];
                row_src[r] = idx[r];
                row_k[r] = kk;
                row_first[r] = (long long)f0;
                row_out[r] = (long long)(lo2[(size_t)k] + i) * (long long)n2 + lo2[(size_t)k];
            }
        }
        if (cusk_gather_rows(e, scr.C.p, (int)n1, idx.data(), (long long)rows2, row_src.data(), row_k.data(), row_first.data(),
                             row_out.data(), (long long)rows2, scr.C2.p, 0, 1) != CUSK_OK)
            engine_die("gather (batch)", e);
    }
    mark("gather2");
    bs.ms_prune = ms_since(t);

    // ---- stage two: Skeleton again on every reduced set, each from its complete graph ----
    if (cusk_run_skeleton_batch(e, scr.C2.p, (int)n2, K, lo2.data(), hi2.data(), in.Th, in.max_level_two, &bs.stage[1]) != CUSK_OK)
        engine_die("Skeleton (stage two, batch)", e);
    mark("stage2");
    for (int l = 0; l < bs.stage[1].levels_run; l++)
    {
        bs.tests[1] += bs.stage[1].tests[l];
        bs.canonical[1] += bs.stage[1].canonical_tests[l];
    }
    bs.ms_stage2 = ms_since(t);

    // ---- reduction (parent_set.cpp:84-175 per block): adjacency, correlations, separating sets of the retained sets ----
    {
        size_t words = 0;
        for (int k = 0; k < K; k++) words += (size_t)(hi2[(size_t)k] - lo2[(size_t)k]) * (size_t)((hi2[(size_t)k] - lo2[(size_t)k] + 63) / 64);
        packed.resize(words);
        if (cusk_result_adj_bits_blocks(e, packed.data()) != CUSK_OK) engine_die("adjacency (stage two, batch)", e);
    }
    mark("bits2");
    std::vector<std::vector<int>> P2((size_t)K);
    size_t rows3 = 0, cells3 = 0;
    {
        size_t pos = 0;
        for (int k = 0; k < K; k++)
        {
            const int kb = hi2[(size_t)k] - lo2[(size_t)k];
            const Bits G2 = block_bits(packed, pos, kb);
            P2[(size_t)k] = subset_variables(G2, kb, kb - (int)p, in.depth);
            Reduced &out = outs[(size_t)kept[(size_t)k]].r;
            out.num_var = P2[(size_t)k].size();
            out.num_phen = p;
            out.max_level = ML;
            out.new_to_old = compose(P2[(size_t)k], &P1[(size_t)k]);
            out.G = gather_adj(G2, P2[(size_t)k]);
            rows3 += P2[(size_t)k].size();
            cells3 += P2[(size_t)k].size() * P2[(size_t)k].size();
            bs.retained += (long long)out.num_markers();
        }
    }
    {
        std::vector<int> idx(rows3), row_src(rows3), row_k(rows3);
        std::vector<long long> row_first(rows3), row_out(rows3);
        std::vector<float> corr(cells3);
        size_t r = 0, cell = 0;
        for (int k = 0; k < K; k++)
        {
            const size_t f0 = r;
            const int kk = (int)P2[(size_t)k].size();
            for (int i = 0; i < kk; i++, r++)
            {
                idx[r] = lo2[(size_t)k] + P2[(size_t)k][(size_t)i];
                row_src[r] = idx[r];
                row_k[r] = kk;
                row_first[r] = (long long)f0;
                row_out[r] = (long long)(cell + (size_t)i * (size_t)kk);
            }
            cell += (size_t)kk * (size_t)kk;
        }
        if (cusk_gather_rows(e, scr.C2.p, (int)n2, idx.data(), (long long)rows3, row_src.data(), row_k.data(), row_first.data(),
                             row_out.data(), (long long)rows3, corr.data(), (long long)cells3, 0) != CUSK_OK)
            engine_die("gather (results, batch)", e);
        cell = 0;
        for (int k = 0; k < K; k++)
        {
            const size_t kk = P2[(size_t)k].size();
            outs[(size_t)kept[(size_t)k]].r.C.assign(corr.begin() + (long)cell, corr.begin() + (long)(cell + kk * kk));
            cell += kk * kk;
        }
    }
    mark("adj_corr");
    // separating sets: the sparse records of the stage-two run (ordered by (x, y), i.e. block after block), reduced as
    // reduce_sepsets does, including the stage-two remap quirk: a member s (an index of the stage-two
    // space) is looked up in a map keyed by the STAGE-ONE index of the retained variables
    {
        // (round 3, second form: the sets stay a sorted list per block -- Reduced::sep_recs -- and the dense k x k x 14 arrays
        // are only built for a caller that asks for them; files and the packed form with_sep = 2 are made from the list)
        for (int k = 0; k < K; k++)
        {
            Reduced &out = outs[(size_t)kept[(size_t)k]].r;
            out.sep_sparse = true;
            out.sep_recs.clear();
        }
        const int *x = nullptr, *y = nullptr, *rs = nullptr;  // engine-owned pinned memory
        const long long cnt = cusk_result_sepsets_view(e, &x, &y, &rs);
        if (cnt < 0) engine_die("sepsets (batch)", e);
        if (cnt > 0)
        {
            int k = 0;
            std::vector<int> pos, o2n;
            int cur = -1;
            for (long long r = 0; r < cnt; r++)
            {
                while (k < K && x[(size_t)r] >= hi2[(size_t)k]) k++;
                if (k >= K || x[(size_t)r] < lo2[(size_t)k]) die("separating-set record outside every block");
                if (cur != k)
                {
                    cur = k;
                    const int kb = hi2[(size_t)k] - lo2[(size_t)k];
                    pos.assign((size_t)kb, -1);
                    o2n.assign((size_t)kb, -1);
                    const std::vector<int> &P = P2[(size_t)k];
                    for (size_t i = 0; i < P.size(); i++)
                    {
                        pos[(size_t)P[i]] = (int)i;
                        const int key = P1[(size_t)k][(size_t)P[i]];
                        if (key < kb) o2n[(size_t)key] = (int)i;
                    }
                }
                const int lx = x[(size_t)r] - lo2[(size_t)k], ly = y[(size_t)r] - lo2[(size_t)k];
                const int ix = pos[(size_t)lx], iy = pos[(size_t)ly];
                if (ix < 0 || iy < 0) continue;
                Reduced &out = outs[(size_t)kept[(size_t)k]].r;
                SepRec q;
                q.ix = ix;
                q.iy = iy;
                q.cnt = 0;
                for (int l = 0; l < ML; l++)
                {
                    const int sv = rs[(size_t)r * ML + (size_t)l];
                    if (sv == -1) continue;
                    const int ls = sv - lo2[(size_t)k];
                    if (pos[(size_t)ls] >= 0) q.s[q.cnt++] = (o2n[(size_t)ls] >= 0) ? o2n[(size_t)ls] : 0;
                }
                out.sep_recs.push_back(q);
            }
        }
        for (int k = 0; k < K; k++) sort_sep_recs(outs[(size_t)kept[(size_t)k]].r);
    }
    mark("sepsets");
    bs.ms_reduce = ms_since(t);
    if (prof) std::fprintf(stderr, "[batchprof]%s\n", plog.c_str());
}

}  // namespace host
