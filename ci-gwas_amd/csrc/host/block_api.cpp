// block_api.cpp -- C ABI of the block driver (include/cusk_hip.h section 3): a set of LD blocks opened once, any block
// run through host/block_pipeline.h (the very code `mps cusk` runs) on the caller's engine.  Plain host C++.
#include <map>
#include <mutex>

#include "block_pipeline.h"

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
        return *s;
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
extern "C" const int *cusk_block_result_sep(const cusk_block_result *r) { return r ? r->r.S.data() : nullptr; }

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
