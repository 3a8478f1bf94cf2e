#!/usr/bin/env python3
"""`ci-gwas.py`-compatible command line for the cusk path on MI355X.

Mirrors the reference's workflow CLI for the two GPU subcommands
(/root/reference/ci-gwas.py:54-61 `prep-bed`, :63-92 `block`, :64-93 `cusk`, :95-253 `cuskss`, handlers :404-456): same
positional / optional arguments, same range checks, same conversion to the positional argv of
the native `mps` program with literal 'NULL' for absent paths, `subprocess.run(check=True)`.
`cuskss-het` and `cuskss-merged` (README.md:65,75 of the reference names them, its CLI does
not define them) are aliases of `cuskss` that insist on the flags that select that mode.

`sepselect` and `orient-v-structs` (ci-gwas.py:303-358, handlers :467-476) run this package's device-backed
mirror of cusk_postprocessing/sepselect.py (ci-gwas_amd/sepselect.py) and write the same files.
`merge-block-outputs` (ci-gwas.py:255-271, :459-464) and the post-step of a merged `cuskss` run (:452-456) use this
package's mirror of the reference's merge module (ci-gwas_amd/merge.py), pinned by files the reference's own code
wrote (tests/golden/merge).  The rest of the downstream (srfci, mvivw) is the reference's own code and consumes the
files written here unchanged.
"""
from __future__ import annotations

import argparse
import os
import subprocess
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
MPS_PATH = os.environ.get("CUSK_MPS_PATH", os.path.join(_HERE, "csrc", "mps"))


class _Parser(argparse.ArgumentParser):
    def error(self, message):
        sys.stderr.write(f"error: {message}\n")
        self.print_help()
        sys.exit(2)


class TypeCheck:
    """ci-gwas.py:27-40"""

    def __init__(self, type_fn, name, min_val=None, max_val=None):
        self._type_fn, self._name, self._min_val, self._max_val = type_fn, name, min_val, max_val

    def __call__(self, val):
        val = self._type_fn(val)
        if self._min_val is not None and val < self._min_val:
            raise argparse.ArgumentTypeError(f"Minimum {self._name} is {self._min_val}")
        if self._max_val is not None and val > self._max_val:
            raise argparse.ArgumentTypeError(f"Maximum {self._name} is {self._max_val}")
        return val


def _add_cusk(sub):
    p = sub.add_parser("cusk", help="Infer skeleton with markers and traits as nodes, using marker data (requires GPU)")
    p.add_argument("block_index", metavar="block-index", type=TypeCheck(int, "block-index", 0, None))
    p.add_argument("blocks", type=str, help="file with genomic block definitions (output of ci-gwas block)")
    p.add_argument("bfiles", type=str, help="filestem of .bed, .bim, .fam fileset")
    p.add_argument("phen", type=str, help="path to standardized phenotype tsv.")
    p.add_argument("alpha", type=TypeCheck(float, "alpha", 0.0, 1.0), default=10**-4)
    p.add_argument("max_level", metavar="max-level", type=TypeCheck(int, "max-level", 0, 14), default=3)
    p.add_argument("max_level_two", metavar="max-level-two", type=TypeCheck(int, "max-level", 0, 14), default=14)
    p.add_argument("max_depth", metavar="max-depth", type=TypeCheck(int, "max-depth", 1, None), default=1)
    p.add_argument("outdir", type=str, default="./")
    p.set_defaults(func=cusk)


def _add_prep(sub):
    """ci-gwas.py:54-61"""
    p = sub.add_parser("prep-bed", help="Prepare PLINK bed file for cusk")
    p.add_argument("bfiles", type=str, help="filestem of .bed, .bim, .fam fileset")
    p.set_defaults(func=prep_bed)


def _add_block(sub):
    """ci-gwas.py:63-92"""
    p = sub.add_parser("block", help="Tile whole-genome LD matrix into block diagonal matrix (requires GPU)")
    p.add_argument("bfiles", type=str, help="filestem of .bed, .bim, .fam fileset")
    p.add_argument("max_block_size", metavar="max-block-size", help="maximum number of markers per block", default=11000,
                   type=TypeCheck(int, "max-block-size", 2, None))
    p.add_argument("device_mem_gb", metavar="device-mem-gb", help="maximum memory available on GPU in GB", default=10,
                   type=TypeCheck(int, "device-mem-gb", 0, None))
    p.add_argument("corr_width", metavar="corr-width", help="width of banded-correlation matrix", default=2000,
                   type=TypeCheck(int, "corr-width", 2, None))
    p.set_defaults(func=block)


def _add_cuskss(sub, name, help_):
    p = sub.add_parser(name, help=help_)
    p.add_argument("--mxm", type=str, default="NULL")
    p.add_argument("--mxp", type=str, default="NULL")
    p.add_argument("--pxp", type=str, required=True)
    p.add_argument("--mxp-se", type=str, default="NULL")
    p.add_argument("--pxp-se", type=str, default="NULL")
    p.add_argument("--block-index", metavar="block-index", type=TypeCheck(int, "block-index", 0, None), default=0)
    p.add_argument("--blockfile", type=str, default="NULL")
    p.add_argument("--marker-indices", metavar="marker-indices", type=str, default="NULL")
    p.add_argument("--alpha", type=TypeCheck(float, "alpha", 0.0, 1.0), required=True)
    p.add_argument("--max-level-one", metavar="max-level", type=TypeCheck(int, "max-level", 0, 14), default=3)
    p.add_argument("--max-level-two", metavar="max-level-two", type=TypeCheck(int, "max-level", 0, 14), default=14)
    p.add_argument("--max-depth", metavar="max-depth", type=TypeCheck(int, "max-depth", 1, None), default=1)
    p.add_argument("--time-index", type=str, default="NULL")
    p.add_argument("--num-samples", metavar="num-samples", type=TypeCheck(int, "num-samples", 1, None), required=True)
    p.add_argument("--outdir", type=str, default="./")
    p.set_defaults(func=cuskss, variant=name)


def _add_merge(sub):
    """ci-gwas.py:255-271"""
    p = sub.add_parser("merge-block-outputs", help="Merge outputs of cusk for all blocks into single files")
    p.add_argument("cusk_output_dir", metavar="cusk-output-dir", type=str, help="output directory of cusk")
    p.add_argument("blockfile", type=str, help="file with genomic block definitions (output of ci-gwas block)")
    p.set_defaults(func=merge_blocks)


def _add_sepselect(sub):
    """ci-gwas.py:303-358"""
    for name, help_, func in (
            ("orient-v-structs", "Orient v-structures using maximal separation sets on merged cusk skeletons.", run_v_struct),
            ("sepselect", "Compute maximal and partial-correlation-minimizing separation sets on merged cusk skeletons",
             run_sepselect)):
        p = sub.add_parser(name, help=help_)
        p.add_argument("cusk_result_stem", metavar="cusk-result-stem", type=str, help="outdir + stem of merged cusk results")
        p.add_argument("alpha", type=TypeCheck(float, "alpha", 0.0, 1.0), default=10**-4,
                       help="significance level for conditional independence tests")
        p.add_argument("num_samples", metavar="num-samples", type=TypeCheck(int, "num-samples", 1, None),
                       help="number of samples used for computing correlations")
        if name == "orient-v-structs":
            p.add_argument("--orientation-prior", metavar="orientation-prior", type=str, default=None,
                           help="matrix of (0, 1) (32 bit integers, binary) of dims (n_trait, n_trait) indicating "
                                "directions to be forced. ")
        p.set_defaults(func=func)


def build_parser() -> argparse.ArgumentParser:
    parser = _Parser(prog="ci-gwas", description="cusk / cuskss steps of CI-GWAS on AMD Instinct MI355X")
    sub = parser.add_subparsers(required=True, title="subcommands")
    _add_prep(sub)
    _add_block(sub)
    _add_cusk(sub)
    _add_cuskss(sub, "cuskss", "Infer skeleton using summary statistic data (requires GPU)")
    _add_cuskss(sub, "cuskss-het", "cuskss with heterogeneous (polychoric/polyserial) correlations: needs --mxp-se/--pxp-se")
    _add_cuskss(sub, "cuskss-merged", "cuskss on the union of markers selected in all blocks: needs --marker-indices")
    _add_merge(sub)
    _add_sepselect(sub)
    return parser


def prep_bed(args):
    """ci-gwas.py:386-387"""
    subprocess.run([MPS_PATH, "prep", args.bfiles], check=True)


def block_argv(args) -> list[str]:
    """ci-gwas.py `block` handler"""
    return [MPS_PATH, "block", args.bfiles, str(args.max_block_size), str(args.device_mem_gb), str(args.corr_width)]


def block(args):
    subprocess.run(block_argv(args), check=True)


def cusk_argv(args) -> list[str]:
    """ci-gwas.py:404-420"""
    return [MPS_PATH, "cusk", args.phen, args.bfiles, args.blocks, str(args.alpha), str(args.max_level),
            str(args.max_level_two), str(args.max_depth), args.outdir, str(args.block_index)]


def cuskss_argv(args) -> list[str]:
    """ci-gwas.py:423-451 (validation :424-429 included)"""
    if args.blockfile == "NULL" and args.marker_indices == "NULL":
        sys.exit("Either blockfile + block index or marker indices into the mxp file have to be provided for cuskss.")
    if sum([args.mxp_se == "NULL", args.pxp_se == "NULL"]) == 1:
        sys.exit("Please provide no or both pxp and mxp standard error files.")
    if sum([args.mxp == "NULL", args.mxm == "NULL"]) == 1:
        sys.exit("Please provide no or both mxp and mxm correlation files.")
    variant = getattr(args, "variant", "cuskss")
    if variant == "cuskss-het" and args.mxp_se == "NULL":
        sys.exit("cuskss-het needs --mxp-se and --pxp-se.")
    if variant == "cuskss-merged" and args.marker_indices == "NULL":
        sys.exit("cuskss-merged needs --marker-indices.")
    return [MPS_PATH, "cuskss", args.mxm, args.mxp, args.mxp_se, args.pxp, args.pxp_se, args.time_index,
            str(args.block_index), args.blockfile, args.marker_indices, str(args.alpha), str(args.max_level_one),
            str(args.max_level_two), str(args.max_depth), str(args.num_samples), args.outdir]


def cusk(args):
    subprocess.run(cusk_argv(args), check=True)


def cuskss(args):
    subprocess.run(cuskss_argv(args), check=True)
    if args.marker_indices != "NULL":
        # ci-gwas.py:452-456: the merged output is rewritten into the sparse merge format (needs
        # <outdir>/merged_blocks.ixs, as in the reference)
        from .merge import reformat_cuskss_merged_output

        reformat_cuskss_merged_output(cusk_dir=args.outdir).write_mm(basepath=f"{args.outdir}/cuskss_merged")


def merge_blocks(args):
    """ci-gwas.py:459-464"""
    from .merge import merge_block_outputs

    out_dir = args.cusk_output_dir if args.cusk_output_dir.endswith("/") else args.cusk_output_dir + "/"
    merge_block_outputs(args.blockfile, out_dir).write_mm(f"{args.cusk_output_dir}/merged_blocks")


def run_sepselect(args):
    """ci-gwas.py:467-470"""
    from .sepselect import sepselect_merged

    merged_cusk = sepselect_merged(args.cusk_result_stem, args.alpha, args.num_samples)
    merged_cusk.to_file(f"{os.path.dirname(args.cusk_result_stem)}/max_sep_min_pc")
    print("Sepselect done.")


def run_v_struct(args):
    """ci-gwas.py:473-476"""
    from .sepselect import orient_v_structures_merged

    merged_cusk = orient_v_structures_merged(args.cusk_result_stem, args.alpha, args.num_samples, args.orientation_prior)
    merged_cusk.to_file(f"{os.path.dirname(args.cusk_result_stem)}/max_sep_min_pc")
    print("Sepselect / v-structs done.")


def main(argv=None):
    args = build_parser().parse_args(argv)
    args.func(args)


if __name__ == "__main__":
    main()
