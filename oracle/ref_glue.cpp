// C glue over the REFERENCE's own host-side C++ (compiled from
// /root/reference/cusk/src/*.cpp where they lie, see oracle/Makefile target
// `ref`).  TEST INFRASTRUCTURE ONLY: the resulting oracle/_ref/libref_host.so
// pins the oracle's numpy restatement of graph reduction, result files and
// loaders.  Nothing of the reference is copied into this repository; the CUDA
// translation units (cuPC-S.cu, hetcor-cuPC-S.cu, corr_*.cu, cli.cpp) and
// cuPC_call_prep.cpp (boost) cannot be built in this image and are not part
// of this library.
#include <mps/blocking.h>
#include <mps/io.h>
#include <mps/marker_summary_stats.h>
#include <mps/marker_trait_summary_stats.h>
#include <mps/parent_set.h>
#include <mps/phen.h>
#include <mps/prep.h>
#include <mps/trait_summary_stats.h>

#include <cstring>
#include <string>
#include <unordered_set>
#include <vector>

extern "C"
{
    // parent_set.cpp:8-53 ; out must hold num_var ints ; returns count (sorted ascending)
    int ref_subset_variables(const int *G, int num_var, int num_markers, int max_depth, int *out)
    {
        std::vector<int> g(G, G + (size_t)num_var * num_var);
        std::unordered_set<int> s = subset_variables(g, num_var, num_markers, max_depth);
        std::vector<int> v = set_to_vec(s);
        std::memcpy(out, v.data(), v.size() * sizeof(int));
        return (int)v.size();
    }

    // parent_set.cpp:84-175 + parent_set.h:42-52 ; writes <base>.{mdim,ixs,adj,corr,sep}
    void ref_reduce_gcs_to_file(
        const int *G, const float *C, const int *S, const int *P, int nP, int num_var, int num_phen,
        int max_level, const int *index_map, const char *base
    )
    {
        size_t nn = (size_t)num_var * num_var;
        std::vector<int> g(G, G + nn);
        std::vector<float> c(C, C + nn);
        std::vector<int> s(S, S + nn * 14);
        std::unordered_set<int> p(P, P + nP);
        ReducedGCS r;
        if (index_map)
        {
            std::vector<int> im(index_map, index_map + num_var);
            r = reduce_gcs(g, c, s, p, num_var, num_phen, max_level, im);
        }
        else
        {
            r = reduce_gcs(g, c, s, p, num_var, num_phen, max_level);
        }
        r.to_file(std::string(base));
    }

    // parent_set.cpp:177-238 + parent_set.h:99-108 ; writes <base>.{mdim,ixs,adj,corr}
    void ref_reduce_gc_to_file(
        const int *G, const float *C, const float *S, const int *P, int nP, int num_var, int num_phen,
        int max_level, const int *index_map, const char *base
    )
    {
        size_t nn = (size_t)num_var * num_var;
        std::vector<int> g(G, G + nn);
        std::vector<float> c(C, C + nn);
        std::vector<float> s(S, S + nn);
        std::unordered_set<int> p(P, P + nP);
        ReducedGC r;
        if (index_map)
        {
            std::vector<int> im(index_map, index_map + num_var);
            r = reduce_gc(g, c, s, p, num_var, num_phen, max_level, im);
        }
        else
        {
            r = reduce_gc(g, c, s, p, num_var, num_phen, max_level);
        }
        r.to_file(std::string(base));
    }

    // marker_summary_stats.cpp:8-24 ; out may be NULL to query the size
    int ref_load_mxm(const char *path, float *out)
    {
        MarkerSummaryStats s{std::string(path)};
        std::vector<float> c = s.get_corrs();
        if (out) std::memcpy(out, c.data(), c.size() * sizeof(float));
        return s.get_num_markers();
    }

    // trait_summary_stats.cpp ; se_path NULL -> (path, sample_size) constructor
    int ref_load_pxp(const char *path, const char *se_path, float sample_size, float *corr, float *ess)
    {
        TraitSummaryStats s = se_path ? TraitSummaryStats(std::string(path), std::string(se_path))
                                      : TraitSummaryStats(std::string(path), sample_size);
        std::vector<float> c = s.get_corrs(), e = s.get_sample_sizes();
        if (corr) std::memcpy(corr, c.data(), c.size() * sizeof(float));
        if (ess) std::memcpy(ess, e.data(), e.size() * sizeof(float));
        return s.get_num_phen();
    }

    // marker_trait_summary_stats.cpp ; rows selected by a block (first/last/global offset) when
    // marker_ixs is NULL, else by the ascending marker index list.  Returns num_markers.
    int ref_load_mxp(
        const char *path, const char *se_path, const int *marker_ixs, int n_ixs, const char *chr,
        int first, int last, int offset, float *corr, float *ess, int *num_phen
    )
    {
        MarkerTraitSummaryStats s;
        if (marker_ixs)
        {
            // the reference reads marker_ixs[num_markers] once past the end; pad with -1
            std::vector<int> ix(marker_ixs, marker_ixs + n_ixs);
            ix.push_back(-1);
            s = se_path ? MarkerTraitSummaryStats(std::string(path), std::string(se_path), ix)
                        : MarkerTraitSummaryStats(std::string(path), ix);
        }
        else
        {
            MarkerBlock b(std::string(chr), first, last, offset);
            s = se_path ? MarkerTraitSummaryStats(std::string(path), std::string(se_path), b)
                        : MarkerTraitSummaryStats(std::string(path), b);
        }
        std::vector<float> c = s.get_corrs(), e = s.get_sample_sizes();
        if (corr) std::memcpy(corr, c.data(), c.size() * sizeof(float));
        if (ess && !e.empty()) std::memcpy(ess, e.data(), e.size() * sizeof(float));
        *num_phen = s.get_num_phen();
        return s.get_num_markers();
    }

    // io.cpp:74-101 ; first/last/offset arrays sized by the caller ; returns number of blocks
    int ref_read_blocks(const char *path, int *first, int *last, int *offset, int cap)
    {
        std::vector<MarkerBlock> b = read_blocks_from_file(std::string(path));
        for (size_t i = 0; i < b.size() && (int)i < cap; i++)
        {
            first[i] = (int)b[i].get_first_marker_ix();
            last[i] = (int)b[i].get_last_marker_ix();
            offset[i] = (int)(b[i].get_first_marker_global_ix() - b[i].get_first_marker_ix());
        }
        return (int)b.size();
    }

    // phen.cpp:27-74 ; data column-major ; returns num_samples
    int ref_load_phen(const char *path, float *data, int *num_phen)
    {
        Phen p = load_phen(std::string(path));
        if (data) std::memcpy(data, p.data.data(), p.data.size() * sizeof(float));
        *num_phen = (int)p.num_phen;
        return (int)p.num_samples;
    }

    // io.cpp:238-249 with BimInfo/BedDims from <stem>.bim/.fam ; returns bytes written
    int ref_read_block_from_bed(const char *stem, const char *chr, int first, int last, unsigned char *out)
    {
        BfilesBase bf{std::string(stem)};
        BedDims dims(bf);
        BimInfo bim(bf.bim());
        MarkerBlock b(std::string(chr), first, last, 0);
        std::vector<unsigned char> v = read_block_from_bed(bf.bed(), b, dims, bim);
        if (out) std::memcpy(out, v.data(), v.size());
        return (int)v.size();
    }

    // blocking.cpp:85-136 ; first/last must hold cap entries ; returns the number of blocks (or -1)
    int ref_block_chr(const float *v, int n, int max_block_size, long long *first, long long *last, int cap)
    {
        std::vector<float> vv(v, v + n);
        std::vector<MarkerBlock> b = block_chr(vv, "1", max_block_size);
        if ((int)b.size() > cap) return -1;
        for (size_t i = 0; i < b.size(); i++)
        {
            first[i] = (long long)b[i].get_first_marker_ix();
            last[i] = (long long)b[i].get_last_marker_ix();
        }
        return (int)b.size();
    }

    // prep.cpp:159-203 (`mps prep`): writes <stem>.dim/.means/.stds/.modes next to the bfiles
    void ref_prep(const char *stem) { prep_bed_no_impute(BfilesBase(std::string(stem))); }

    // blocking.cpp:13-35
    void ref_hanning_smoothing(const float *v, int n, int window_size, double *out)
    {
        std::vector<float> vv(v, v + n);
        std::vector<double> r = hanning_smoothing(vv, window_size);
        std::memcpy(out, r.data(), sizeof(double) * r.size());
    }
}
