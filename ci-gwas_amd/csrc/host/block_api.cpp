// block_api.cpp -- C ABI of the block driver (include/cusk_hip.h section 3): a set of LD blocks opened once, any block
// run through host/block_pipeline.h (the very code `mps cusk` runs) on the caller's engine.  Plain host C++.
#include <map>
#include <mutex>
#include <thread>

#include "batch_pipeline.h"
#include "merge.h"

using namespace host;

struct cusk_blockset
{
    CuskInputs in;
    // device scratch (the block's matrix, the stage-two matrix) per engine, reused from block to block and released
    // with the block set
    std::mutex mu;
    std::map<cusk_engine *, std::unique_ptr<BlockScratch>> scratch;
    std::map<int, StagedInputs> staged;  // by device ordinal
    const StagedInputs *staged_of(cusk_engine *e)
    {
        std::lock_guard<std::mutex> lock(mu);
        auto it = staged.find(cusk_engine_device(e));
        return it == staged.end() ? nullptr : &it->second;
    }
    ~cusk_blockset()
    {
        for (auto &kv : staged) kv.second.release();
    }
    BlockScratch &scratch_of(cusk_engine *e)
    {
        std::lock_guard<std::mutex> lock(mu);
        auto &s = scratch[e];
        if (!s) s.reset(new BlockScratch());
        // an entry left behind by an engine that was destroyed while a build was pending (a new engine can get the same
        // address): the engine itself knows whether a build is in flight
        if (s->pending >= 0 && !cusk_corr_build_pending(e)) s->pending = -1;
        return *s;
    }
    std::map<cusk_engine *, std::unique_ptr<BatchScratch>> batch_scratch;
    BatchScratch &batch_scratch_of(cusk_engine *e)
    {
        std::lock_guard<std::mutex> lock(mu);
        auto &s = batch_scratch[e];
        if (!s) s.reset(new BatchScratch());
        return *s;
    }
    void release_engine(cusk_engine *e)
    {
        std::lock_guard<std::mutex> lock(mu);
        scratch.erase(e);
        batch_scratch.erase(e);
    }
};

struct cusk_block_result
{
    Reduced r;
    std::string stem;
};

namespace {
thread_local std::string g_err;

void copy_err(const std::string &msg, char *err, size_t err_len)
{
    g_err = msg;
    if (err && err_len)
    {
        std::strncpy(err, msg.c_str(), err_len - 1);
        err[err_len - 1] = 0;
    }
}
}  // namespace

extern "C" int cusk_blockset_open(cusk_blockset **out, const char *phen_path, const char *bfiles,
                                  const char *blocks_path, float alpha, int max_level, int max_level_two, int depth,
                                  char *err, size_t err_len)
{
    if (!out || !phen_path || !bfiles || !blocks_path) return CUSK_ERR_ARG;
    *out = nullptr;
    std::unique_ptr<cusk_blockset> bs(new cusk_blockset());
    bs->in.phen_path = phen_path;
    bs->in.bfiles = bfiles;
    bs->in.block_path = blocks_path;
    bs->in.alpha = alpha;
    bs->in.max_level = max_level;
    bs->in.max_level_two = max_level_two;
    bs->in.depth = depth;
    try
    {
        bs->in.load(nullptr);
        bs->in.load_all_marker_stats();
    }
    catch (const std::exception &ex)
    {
        copy_err(ex.what(), err, err_len);
        return CUSK_ERR_ARG;
    }
    *out = bs.release();
    return CUSK_OK;
}

extern "C" void cusk_blockset_close(cusk_blockset *bs) { delete bs; }
extern "C" int cusk_blockset_num_blocks(const cusk_blockset *bs) { return bs ? (int)bs->in.blocks.size() : 0; }
extern "C" long long cusk_blockset_num_samples(const cusk_blockset *bs) { return bs ? (long long)bs->in.dims.num_samples : 0; }
extern "C" int cusk_blockset_num_phen(const cusk_blockset *bs) { return bs ? (int)bs->in.phen.num_phen : 0; }

extern "C" long long cusk_blockset_block_markers(const cusk_blockset *bs, int i)
{
    if (!bs || i < 0 || (size_t)i >= bs->in.blocks.size()) return -1;
    return (long long)bs->in.blocks[i].size();
}

extern "C" int cusk_blockset_block_stem(const cusk_blockset *bs, int i, char *stem, size_t stem_len)
{
    if (!bs || !stem || !stem_len || i < 0 || (size_t)i >= bs->in.blocks.size()) return CUSK_ERR_ARG;
    const std::string s = bs->in.blocks[i].file_stem();
    if (s.size() + 1 > stem_len) return CUSK_ERR_ARG;
    std::memcpy(stem, s.c_str(), s.size() + 1);
    return CUSK_OK;
}

extern "C" int cusk_blockset_stage(cusk_blockset *bs, cusk_engine *e)
{
    if (!bs || !e) return CUSK_ERR_ARG;
    if (bs->staged_of(e)) return CUSK_OK;
    if (cusk_engine_bind_thread(e) != CUSK_OK) return CUSK_ERR_HIP;
    const CuskInputs &in = bs->in;
    const size_t bed_bytes = in.bed.size > 3 ? in.bed.size - 3 : 0, pad = 4096;
    StagedInputs st;
    st.bed = static_cast<unsigned char *>(cusk_dev_alloc(bed_bytes + pad));
    st.phen = static_cast<float *>(cusk_dev_alloc(sizeof(float) * in.phen.data.size() + pad));
    st.means = static_cast<float *>(cusk_dev_alloc(sizeof(float) * in.means_all.size() + pad));
    st.stds = static_cast<float *>(cusk_dev_alloc(sizeof(float) * in.stds_all.size() + pad));
    bool ok = st.bed && st.phen && st.means && st.stds;
    ok = ok && cusk_dev_upload(st.bed, in.bed.data + 3, bed_bytes) == CUSK_OK;
    ok = ok && cusk_dev_upload(st.phen, in.phen.data.data(), sizeof(float) * in.phen.data.size()) == CUSK_OK;
    ok = ok && cusk_dev_upload(st.means, in.means_all.data(), sizeof(float) * in.means_all.size()) == CUSK_OK;
    ok = ok && cusk_dev_upload(st.stds, in.stds_all.data(), sizeof(float) * in.stds_all.size()) == CUSK_OK;
    if (!ok)
    {
        st.release();
        copy_err("staging the block set's inputs on the device failed (not enough device memory?)", nullptr, 0);
        return CUSK_ERR_HIP;
    }
    std::lock_guard<std::mutex> lock(bs->mu);
    bs->staged[cusk_engine_device(e)] = st;
    return CUSK_OK;
}

extern "C" int cusk_blockset_run_block_next(cusk_blockset *bs, cusk_engine *e, int block_index, int next_index,
                                            cusk_block_result **out, cusk_block_stats *stats);

extern "C" int cusk_blockset_run_block(cusk_blockset *bs, cusk_engine *e, int block_index, cusk_block_result **out,
                                       cusk_block_stats *stats)
{
    return cusk_blockset_run_block_next(bs, e, block_index, -1, out, stats);
}

// next_index >= 0: the block this engine runs next -- its correlation matrix is built beside this block's sweeps (staged
// inputs only; a different block may still be asked for next, the build is then dropped)
extern "C" int cusk_blockset_run_block_next(cusk_blockset *bs, cusk_engine *e, int block_index, int next_index,
                                            cusk_block_result **out, cusk_block_stats *stats)
{
    if (!bs || !e || !out) return CUSK_ERR_ARG;
    *out = nullptr;
    BlockScratch &scratch = bs->scratch_of(e);
    std::unique_ptr<cusk_block_result> res(new cusk_block_result());
    BlockStats st;
    try
    {
        const bool kept = run_cusk_block(e, bs->in, block_index, scratch, res->r, res->stem, st, nullptr, bs->staged_of(e), next_index);
        if (stats)
        {
            stats->skipped = st.skipped;
            stats->num_sig = st.num_sig;
            stats->markers = st.markers;
            stats->retained = st.retained;
            stats->tests[0] = st.tests[0];
            stats->tests[1] = st.tests[1];
            stats->ms_inputs = st.ms_inputs;
            stats->ms_corr = st.ms_corr;
            stats->ms_stage1 = st.ms_stage1;
            stats->ms_prune = st.ms_prune;
            stats->ms_stage2 = st.ms_stage2;
            stats->ms_reduce = st.ms_reduce;
            stats->stage[0] = st.stage[0];
            stats->stage[1] = st.stage[1];
        }
        if (kept) *out = res.release();
    }
    catch (const EngineError &ex)
    {
        copy_err(ex.what(), nullptr, 0);
        return CUSK_ERR_HIP;
    }
    catch (const std::exception &ex)
    {
        copy_err(ex.what(), nullptr, 0);
        return CUSK_ERR_ARG;
    }
    return CUSK_OK;
}

extern "C" const char *cusk_blockset_last_error(void) { return g_err.c_str(); }

extern "C" void cusk_blockset_release_engine(cusk_blockset *bs, cusk_engine *e)
{
    if (!bs || !e) return;
    if (cusk_corr_build_pending(e)) (void)cusk_corr_build_end(e, nullptr);  // nothing may still write into the scratch
    (void)cusk_engine_bind_thread(e);
    bs->release_engine(e);
}

// ---- many blocks per device run (host/batch_pipeline.h) ----
struct cusk_batch_result
{
    std::vector<cusk_block_result> blocks;
    std::vector<int> index;
};

extern "C" int cusk_blockset_run_batch(cusk_blockset *bs, cusk_engine *e, const int *block_indices, int nblocks,
                                       cusk_batch_result **out, cusk_batch_stats *stats)
{
    if (!bs || !e || !out || !block_indices || nblocks < 0) return CUSK_ERR_ARG;
    *out = nullptr;
    if (!bs->staged_of(e) && cusk_blockset_stage(bs, e) != CUSK_OK) return CUSK_ERR_HIP;
    const StagedInputs *staged = bs->staged_of(e);
    BatchScratch &scratch = bs->batch_scratch_of(e);
    std::unique_ptr<cusk_batch_result> res(new cusk_batch_result());
    try
    {
        std::vector<BatchBlockOut> outs;
        BatchStats st;
        run_cusk_batch(e, bs->in, *staged, std::vector<int>(block_indices, block_indices + nblocks), scratch, outs, st);
        for (BatchBlockOut &o : outs)
        {
            if (o.skipped) continue;
            res->blocks.emplace_back();
            res->blocks.back().r = std::move(o.r);
            res->blocks.back().stem = std::move(o.stem);
            res->index.push_back(o.block_index);
        }
        if (stats)
        {
            stats->blocks = st.blocks;
            stats->skipped = st.skipped;
            stats->markers = st.markers;
            stats->retained = st.retained;
            stats->vars_stage1 = st.vars_stage1;
            stats->vars_stage2 = st.vars_stage2;
            for (int k = 0; k < 2; k++)
            {
                stats->tests[k] = st.tests[k];
                stats->canonical[k] = st.canonical[k];
                stats->stage[k] = st.stage[k];
            }
            stats->ms_corr = st.ms_corr;
            stats->ms_stage1 = st.ms_stage1;
            stats->ms_prune = st.ms_prune;
            stats->ms_stage2 = st.ms_stage2;
            stats->ms_reduce = st.ms_reduce;
        }
        *out = res.release();
    }
    catch (const EngineError &ex)
    {
        copy_err(ex.what(), nullptr, 0);
        return CUSK_ERR_HIP;
    }
    catch (const std::exception &ex)
    {
        copy_err(ex.what(), nullptr, 0);
        return CUSK_ERR_ARG;
    }
    return CUSK_OK;
}

extern "C" int cusk_batch_result_count(const cusk_batch_result *r) { return r ? (int)r->blocks.size() : 0; }
extern "C" int cusk_batch_result_block_index(const cusk_batch_result *r, int i)
{
    return (r && i >= 0 && (size_t)i < r->index.size()) ? r->index[(size_t)i] : -1;
}
extern "C" const cusk_block_result *cusk_batch_result_block(const cusk_batch_result *r, int i)
{
    return (r && i >= 0 && (size_t)i < r->blocks.size()) ? &r->blocks[(size_t)i] : nullptr;
}
extern "C" void cusk_batch_result_free(cusk_batch_result *r) { delete r; }

// five files per block, a dozen microseconds each: a few threads share the blocks
static void write_many(const std::vector<const cusk_block_result *> &items, const std::string &outdir, bool with_sep)
{
    const size_t nb = items.size();
    const unsigned nt = (unsigned)std::min<size_t>(8, std::max<size_t>(1, nb / 4));
    if (nt <= 1)
    {
        for (const cusk_block_result *b : items) write_reduced(b->r, make_path(outdir, b->stem, ""), with_sep);
        return;
    }
    std::vector<std::thread> th;
    std::vector<std::string> errs(nt);
    for (unsigned t = 0; t < nt; t++)
        th.emplace_back([&, t]() {
            try
            {
                for (size_t i = t; i < nb; i += nt) write_reduced(items[i]->r, make_path(outdir, items[i]->stem, ""), with_sep);
            }
            catch (const std::exception &ex)
            {
                errs[t] = ex.what();
            }
        });
    for (auto &x : th) x.join();
    for (const std::string &m : errs)
        if (!m.empty()) throw std::runtime_error(m);
}

extern "C" int cusk_batch_result_write(const cusk_batch_result *r, const char *outdir)
{
    if (!r || !outdir) return CUSK_ERR_ARG;
    try
    {
        check_path(outdir);
        std::vector<const cusk_block_result *> items;
        for (const cusk_block_result &b : r->blocks) items.push_back(&b);
        write_many(items, outdir, true);
    }
    catch (const std::exception &ex)
    {
        copy_err(ex.what(), nullptr, 0);
        return CUSK_ERR_ARG;
    }
    return CUSK_OK;
}

// with_sep: 0 none, 1 the dense num_var^2 x max_level array, 2 the list (count, then SepRec after SepRec) -- results that
// hold no list go out dense whatever was asked for (the head says which)
static int sep_form_of(const cusk_block_result &b, int with_sep) { return (with_sep == 2 && !b.r.sep_sparse) ? 1 : with_sep; }
static size_t packed_bytes_of(const cusk_block_result &b, int with_sep)
{
    const size_t k = b.r.num_var;
    const int form = sep_form_of(b, with_sep);
    const size_t sep = form == 1 ? 4 * k * k * b.r.max_level : (form == 2 ? 4 + sizeof(SepRec) * b.r.sep_recs.size() : 0);
    return 24 + b.stem.size() + 4 * (k + 2 * k * k) + sep;
}

extern "C" size_t cusk_batch_result_packed_bytes_ex(const cusk_batch_result *r, int with_sep)
{
    size_t t = 0;
    if (r)
        for (const cusk_block_result &b : r->blocks) t += packed_bytes_of(b, with_sep);
    return t;
}
extern "C" size_t cusk_batch_result_packed_bytes(const cusk_batch_result *r) { return cusk_batch_result_packed_bytes_ex(r, 1); }

extern "C" int cusk_batch_result_pack_ex(const cusk_batch_result *r, void *buf, size_t bytes, int with_sep);
extern "C" int cusk_batch_result_pack(const cusk_batch_result *r, void *buf, size_t bytes) { return cusk_batch_result_pack_ex(r, buf, bytes, 1); }

extern "C" int cusk_batch_result_pack_ex(const cusk_batch_result *r, void *buf, size_t bytes, int with_sep)
{
    if (!r || (!buf && bytes)) return CUSK_ERR_ARG;
    if (bytes < cusk_batch_result_packed_bytes_ex(r, with_sep)) return CUSK_ERR_ARG;
    char *p = static_cast<char *>(buf);
    for (size_t i = 0; i < r->blocks.size(); i++)
    {
        const cusk_block_result &b = r->blocks[i];
        const int form = sep_form_of(b, with_sep);
        const int head[6] = {r->index[i], (int)b.r.num_var, (int)b.r.num_phen, (int)b.r.max_level, form, (int)b.stem.size()};
        std::memcpy(p, head, 24);
        p += 24;
        std::memcpy(p, b.stem.data(), b.stem.size());
        p += b.stem.size();
        auto put = [&](const void *src, size_t n) {
            std::memcpy(p, src, n);
            p += n;
        };
        put(b.r.new_to_old.data(), 4 * b.r.new_to_old.size());
        put(b.r.G.data(), 4 * b.r.G.size());
        put(b.r.C.data(), 4 * b.r.C.size());
        if (form == 1)
            put(b.r.dense_sep().data(), 4 * b.r.dense_sep().size());
        else if (form == 2)
        {
            const int nrec = (int)b.r.sep_recs.size();
            put(&nrec, 4);
            put(b.r.sep_recs.data(), sizeof(SepRec) * b.r.sep_recs.size());
        }
    }
    return CUSK_OK;
}

// `merge-block-outputs` (merge_blocks.py:361-395) on the packed results of a whole job: the merged sparse skeleton files
// <basepath>_sam.mtx, _scm.mtx, .mdim, .ixs, without reading the per-block files back (host/merge.h)
extern "C" int cusk_merge_packed(const char *blockfile, const void *buf, size_t bytes, const char *basepath)
{
    if (!blockfile || (!buf && bytes) || !basepath) return CUSK_ERR_ARG;
    try
    {
        const std::vector<Block> listed = read_blocks(blockfile);
        std::map<std::string, size_t> where;  // stem -> line of the .blocks file
        for (size_t i = 0; i < listed.size(); i++) where[listed[i].file_stem()] = i;
        std::vector<MergeInput> in(listed.size());
        std::vector<size_t> sizes(listed.size());
        for (size_t i = 0; i < listed.size(); i++) sizes[i] = listed[i].size();
        // (int / float arrays inside the byte string are 4-byte aligned only if the stems' lengths allow: copy them out)
        std::vector<std::vector<int>> ixs(listed.size()), adj(listed.size());
        std::vector<std::vector<float>> corr(listed.size());
        const char *p = static_cast<const char *>(buf), *end = p + bytes;
        while (p < end)
        {
            if (end - p < 24) throw std::runtime_error("truncated packed results");
            int head[6];
            std::memcpy(head, p, 24);
            p += 24;
            const size_t k = (size_t)head[1], ml = (size_t)head[3], ns = (size_t)head[5];
            const int form = head[4];  // separating sets: 0 none, 1 dense, 2 list (skipped here either way)
            if (form < 0 || form > 2) throw std::runtime_error("packed results: unknown separating-set form");
            if ((size_t)(end - p) < ns + 4 * (k + 2 * k * k + (form == 1 ? k * k * ml : 0)) + (form == 2 ? 4 : 0))
                throw std::runtime_error("truncated packed results");
            const std::string stem(p, ns);
            p += ns;
            auto it = where.find(stem);
            if (it == where.end()) throw std::runtime_error("result of a block that is not in the block file: " + stem);
            const size_t at = it->second;
            ixs[at].resize(k);
            adj[at].resize(k * k);
            corr[at].resize(k * k);
            std::memcpy(ixs[at].data(), p, 4 * k);
            p += 4 * k;
            std::memcpy(adj[at].data(), p, 4 * k * k);
            p += 4 * k * k;
            std::memcpy(corr[at].data(), p, 4 * k * k);
            p += 4 * k * k;
            if (form == 1)
                p += 4 * k * k * ml;
            else if (form == 2)
            {
                int nrec = 0;
                std::memcpy(&nrec, p, 4);
                p += 4;
                if (nrec < 0 || (size_t)(end - p) < sizeof(SepRec) * (size_t)nrec) throw std::runtime_error("truncated packed results");
                p += sizeof(SepRec) * (size_t)nrec;
            }
            in[at].present = true;
            in[at].num_var = k;
            in[at].num_phen = (size_t)head[2];
            in[at].max_level = ml;
            in[at].ixs = ixs[at].data();
            in[at].adj = adj[at].data();
            in[at].corr = corr[at].data();
        }
        const auto tm0 = std::chrono::steady_clock::now();
        merge_blocks_to_files(in, sizes, basepath);
        if (std::getenv("CUSK_BATCH_PROF"))
            std::fprintf(stderr, "[mergeprof] %zu blocks, %zu bytes: total %.0f us\n", listed.size(), bytes,
                         std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - tm0).count());
    }
    catch (const std::exception &ex)
    {
        copy_err(ex.what(), nullptr, 0);
        return CUSK_ERR_ARG;
    }
    return CUSK_OK;
}

extern "C" int cusk_packed_results_write(const void *buf, size_t bytes, const char *outdir, int *blocks_written)
{
    if ((!buf && bytes) || !outdir) return CUSK_ERR_ARG;
    if (blocks_written) *blocks_written = 0;
    try
    {
        check_path(outdir);
        const auto t0 = std::chrono::steady_clock::now();
        const char *p = static_cast<const char *>(buf), *end = p + bytes;
        std::vector<std::unique_ptr<cusk_block_result>> parsed;
        bool all_sep = true;
        while (p < end)
        {
            if (end - p < 24) throw std::runtime_error("truncated packed results");
            int head[6];
            std::memcpy(head, p, 24);
            p += 24;
            const size_t k = (size_t)head[1], ml = (size_t)head[3], ns = (size_t)head[5];
            const int form = head[4];
            const bool has_sep = form != 0;
            if (form < 0 || form > 2) throw std::runtime_error("packed results: unknown separating-set form");
            const size_t need = ns + 4 * (k + 2 * k * k + (form == 1 ? k * k * ml : 0)) + (form == 2 ? 4 : 0);
            if ((size_t)(end - p) < need) throw std::runtime_error("truncated packed results");
            parsed.emplace_back(new cusk_block_result());
            Reduced &r = parsed.back()->r;
            parsed.back()->stem.assign(p, ns);
            p += ns;
            r.num_var = k;
            r.num_phen = (size_t)head[2];
            r.max_level = ml;
            auto take = [&](auto &vec, size_t n) {
                vec.resize(n);
                std::memcpy(vec.data(), p, 4 * n);
                p += 4 * n;
            };
            take(r.new_to_old, k);
            take(r.G, k * k);
            take(r.C, k * k);
            if (form == 1)
                take(r.S, k * k * ml);
            else if (form == 2)
            {
                int nrec = 0;
                std::memcpy(&nrec, p, 4);
                p += 4;
                if (nrec < 0 || (size_t)(end - p) < sizeof(SepRec) * (size_t)nrec) throw std::runtime_error("truncated packed results");
                r.sep_recs.resize((size_t)nrec);
                std::memcpy(r.sep_recs.data(), p, sizeof(SepRec) * (size_t)nrec);
                p += sizeof(SepRec) * (size_t)nrec;
                for (const SepRec &q : r.sep_recs)
                    if (q.ix < 0 || q.iy < 0 || (size_t)q.ix >= k || (size_t)q.iy >= k || q.cnt < 0 || (size_t)q.cnt > ml || q.cnt > ML)
                        throw std::runtime_error("packed results: separating set outside the block");
                r.sep_sparse = true;
                sort_sep_recs(r);
            }
            all_sep = all_sep && has_sep;
            if (blocks_written) (*blocks_written)++;
        }
        std::vector<const cusk_block_result *> items;
        for (auto &b : parsed) items.push_back(b.get());
        const auto t1 = std::chrono::steady_clock::now();
        if (all_sep)
            write_many(items, outdir, true);
        else
            for (const cusk_block_result *b : items) write_reduced(b->r, make_path(outdir, b->stem, ""), b->r.has_sep());
        if (std::getenv("CUSK_BATCH_PROF"))
            std::fprintf(stderr, "[writeprof] %zu blocks, %zu bytes: parse %.0f us, write %.0f us\n", items.size(), bytes,
                         std::chrono::duration<double, std::micro>(t1 - t0).count(),
                         std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t1).count());
    }
    catch (const std::exception &ex)
    {
        copy_err(ex.what(), nullptr, 0);
        return CUSK_ERR_ARG;
    }
    return CUSK_OK;
}

extern "C" void cusk_block_result_dims(const cusk_block_result *r, long long *num_var, long long *num_phen, long long *max_level)
{
    if (num_var) *num_var = r ? (long long)r->r.num_var : 0;
    if (num_phen) *num_phen = r ? (long long)r->r.num_phen : 0;
    if (max_level) *max_level = r ? (long long)r->r.max_level : 0;
}
extern "C" const char *cusk_block_result_stem(const cusk_block_result *r) { return r ? r->stem.c_str() : ""; }
extern "C" const int *cusk_block_result_ixs(const cusk_block_result *r) { return r ? r->r.new_to_old.data() : nullptr; }
extern "C" const int *cusk_block_result_adj(const cusk_block_result *r) { return r ? r->r.G.data() : nullptr; }
extern "C" const float *cusk_block_result_corr(const cusk_block_result *r) { return r ? r->r.C.data() : nullptr; }
extern "C" const int *cusk_block_result_sep(const cusk_block_result *r) { return r ? r->r.dense_sep().data() : nullptr; }

extern "C" int cusk_block_result_write(const cusk_block_result *r, const char *outdir)
{
    if (!r || !outdir) return CUSK_ERR_ARG;
    try
    {
        check_path(outdir);
        write_reduced(r->r, make_path(outdir, r->stem, ""), true);
    }
    catch (const std::exception &ex)
    {
        copy_err(ex.what(), nullptr, 0);
        return CUSK_ERR_ARG;
    }
    return CUSK_OK;
}

extern "C" void cusk_block_result_free(cusk_block_result *r) { delete r; }
