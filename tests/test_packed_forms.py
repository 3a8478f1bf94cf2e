"""CPU: the packed result forms of include/cusk_hip.h (cusk_batch_result_pack_ex) through the library's host-side writer,
cusk_packed_results_write -- no device involved.  Form 1 carries the dense num_var^2 x max_level .sep array (the layout of
shard.BlockResult.pack), form 2 the separating sets as a list that the writer streams into the same file."""
import os

import numpy as np

ML = 14


def _block(rng, index, k, p, stem):
    ixs = rng.permutation(1000)[:k].astype(np.int32)
    adj = (rng.random((k, k)) < 0.2).astype(np.int32)
    corr = rng.standard_normal((k, k)).astype(np.float32)
    cells = rng.permutation(k * k)[: max(1, k * k // 7)]
    recs = []
    for c in np.sort(cells):
        cnt = int(rng.integers(0, ML + 1))
        recs.append((int(c // k), int(c % k), cnt, rng.integers(0, k, ML).astype(np.int32)))
    dense = np.full((k, k, ML), -1, np.int32)
    for ix, iy, cnt, s in recs:
        dense[ix, iy, :cnt] = s[:cnt]
    return dict(index=index, k=k, p=p, stem=stem, ixs=ixs, adj=adj, corr=corr, recs=recs, dense=dense)


def _pack(b, form):
    head = np.array([b["index"], b["k"], b["p"], ML, form, len(b["stem"])], np.int32)
    parts = [head.view(np.uint8), np.frombuffer(b["stem"].encode(), np.uint8), b["ixs"].view(np.uint8),
             b["adj"].reshape(-1).view(np.uint8), b["corr"].reshape(-1).view(np.uint8)]
    if form == 1:
        parts.append(b["dense"].reshape(-1).view(np.uint8))
    elif form == 2:
        parts.append(np.array([len(b["recs"])], np.int32).view(np.uint8))
        for ix, iy, cnt, s in b["recs"]:
            parts.append(np.concatenate([np.array([ix, iy, cnt], np.int32), s]).view(np.uint8))
    return np.concatenate(parts)


def test_list_and_dense_forms_write_the_same_files(tmp_path):
    from cigwas_amd import run_blocks as rb

    rng = np.random.default_rng(5)
    # one block whose array is larger than the writer's 1 MB piece (k^2 x 14 x 4 B = 2.9 MB), small ones, an empty list
    blocks = [_block(rng, 3, 230, 4, "1_0_99"), _block(rng, 0, 7, 2, "1_100_120"), _block(rng, 9, 40, 5, "2_5_60")]
    blocks[1]["recs"], blocks[1]["dense"] = [], np.full((7, 7, ML), -1, np.int32)
    outs = []
    for form in (1, 2):
        out = tmp_path / f"f{form}"
        out.mkdir()
        assert rb.write_packed(np.concatenate([_pack(b, form) for b in blocks]), str(out)) == 3
        outs.append(out)
    files = sorted(os.listdir(outs[0]))
    assert files == sorted(os.listdir(outs[1])) and len(files) == 15
    for f in files:
        assert open(outs[0] / f, "rb").read() == open(outs[1] / f, "rb").read(), f
    for b in blocks:
        assert np.array_equal(np.fromfile(outs[1] / (b["stem"] + ".sep"), np.int32), b["dense"].reshape(-1))
        assert np.array_equal(np.fromfile(outs[1] / (b["stem"] + ".adj"), np.int32), b["adj"].reshape(-1))
        assert open(outs[1] / (b["stem"] + ".mdim")).read() == f"{b['k']}\t{b['p']}\t{ML}\n"
    # mixed forms in one byte string, and records that arrive out of order
    b = blocks[2]
    shuffled = dict(b, recs=[b["recs"][i] for i in rng.permutation(len(b["recs"]))])
    out = tmp_path / "mixed"
    out.mkdir()
    assert rb.write_packed(np.concatenate([_pack(blocks[0], 1), _pack(shuffled, 2)]), str(out)) == 2
    assert open(out / "2_5_60.sep", "rb").read() == open(outs[0] / "2_5_60.sep", "rb").read()


def test_malformed_list_form_is_refused(tmp_path):
    import pytest

    from cigwas_amd import run_blocks as rb

    rng = np.random.default_rng(6)
    b = _block(rng, 1, 12, 3, "1_0_9")
    good = _pack(b, 2)
    with pytest.raises(RuntimeError):
        rb.write_packed(good[:-40], str(tmp_path))  # truncated record list
    bad = dict(b, recs=[(12, 0, 1, np.zeros(ML, np.int32))])  # row outside the block
    with pytest.raises(RuntimeError):
        rb.write_packed(_pack(bad, 2), str(tmp_path))
    head3 = _pack(b, 2).copy()
    head3[16:20] = np.array([3], np.int32).view(np.uint8)  # unknown form
    with pytest.raises(RuntimeError):
        rb.write_packed(head3, str(tmp_path))
