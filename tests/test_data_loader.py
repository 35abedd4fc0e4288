"""Native libsvm reader (arm-net_amd/data_loader.py + csrc/libsvm_reader.cpp) against tensors produced by the
reference's own LibsvmDataset (data_loader.py:12-47) on tests/golden/libsvm_small.libsvm — 60 real Frappe rows,
40 synthetic rows with assorted number spellings and 6 malformed lines the reference skips."""
import os
import types

import numpy as np
import torch

from golden_util import GOLDEN


def _expected():
    return np.load(os.path.join(GOLDEN, "libsvm_small_expected.npz"))


def test_libsvm_dataset_matches_reference_loader(capsys):
    from data_loader import LibsvmDataset
    exp = _expected()
    ds = LibsvmDataset(os.path.join(GOLDEN, "libsvm_small.libsvm"), 10)
    n = int(exp["nsamples"])
    assert ds.nsamples == n == len(ds) and ds.nskipped == int(exp["nlines"]) - n
    assert ds.feat_id.dtype == torch.int64 and ds.feat_value.dtype == torch.float32
    np.testing.assert_array_equal(ds.feat_id[:n].numpy(), exp["ids"])
    np.testing.assert_array_equal(ds.feat_value[:n].numpy(), exp["vals"])       # bit exact: same strtod rounding
    np.testing.assert_array_equal(ds.y[:n].numpy(), exp["y"])
    item = ds[3]
    assert set(item) == {"id", "value", "y"} and item["id"].shape == (10,)
    assert f"# {n} data samples loaded" in capsys.readouterr().out


def test_thread_count_does_not_change_the_result():
    from data_loader import LibsvmDataset
    a = LibsvmDataset(os.path.join(GOLDEN, "libsvm_small.libsvm"), 10, nthreads=1)
    b = LibsvmDataset(os.path.join(GOLDEN, "libsvm_small.libsvm"), 10, nthreads=7)
    assert a.nsamples == b.nsamples
    assert torch.equal(a.feat_id[: a.nsamples], b.feat_id[: b.nsamples])
    assert torch.equal(a.feat_value[: a.nsamples], b.feat_value[: b.nsamples])


def test_libsvm_dataloader_surface(tmp_path):
    from data_loader import libsvm_dataloader
    src = open(os.path.join(GOLDEN, "libsvm_small.libsvm")).read()
    d = tmp_path / "toy"
    d.mkdir()
    for name in ("train.libsvm", "valid.libsvm", "test.libsvm"):
        (d / name).write_text(src)
    args = types.SimpleNamespace(data_dir=str(tmp_path) + "/", dataset="toy", nfield=10, batch_size=32, workers=0)
    tr, va, te = libsvm_dataloader(args)
    batch = next(iter(va))
    assert batch["id"].shape == (32, 10) and batch["value"].dtype == torch.float32 and batch["y"].shape == (32,)
    assert len(tr.dataset) == len(te.dataset) == int(_expected()["nsamples"])


def test_binary_cache_round_trip_and_staleness(tmp_path, capsys):
    """SURVEY §8f-3's binary pre-tokenised format: LibsvmDataset(cache=...) writes one binary file beside the parse, a second
    construction reads it back bit for bit without touching the text parser, and a cache that no longer matches the text
    file (different size / mtime / nfields, truncated, foreign bytes) is ignored and rewritten"""
    import shutil
    import data_loader
    from data_loader import LibsvmDataset
    src = tmp_path / "tr.libsvm"
    shutil.copy(os.path.join(GOLDEN, "libsvm_small.libsvm"), src)
    cache = tmp_path / "tr.bin"
    a = LibsvmDataset(str(src), 10, cache=str(cache))
    assert cache.exists()
    capsys.readouterr()
    real_lib = data_loader._lib
    data_loader._lib = lambda: (_ for _ in ()).throw(AssertionError("the text parser must not run on a cache hit"))
    try:
        b = LibsvmDataset(str(src), 10, cache=str(cache))
    finally:
        data_loader._lib = real_lib
    assert "binary cache" in capsys.readouterr().out
    assert (b.nsamples, b.nskipped) == (a.nsamples, a.nskipped)
    for x, y in ((a.feat_id, b.feat_id), (a.feat_value, b.feat_value), (a.y, b.y)):
        assert x.dtype == y.dtype and torch.equal(x[: a.nsamples], y[: a.nsamples])
    # default name, and the loader surface still works on cached data
    c = LibsvmDataset(str(src), 10, cache=True)
    assert os.path.exists(str(src) + ".armnet.bin") and c.nsamples == a.nsamples
    # stale: the text changed (one more valid line) -> re-parsed, cache rewritten
    with open(src, "a") as f:
        f.write("1 " + " ".join(f"{i}:1" for i in range(10)) + "\n")
    d = LibsvmDataset(str(src), 10, cache=str(cache))
    assert d.nsamples == a.nsamples + 1
    e = LibsvmDataset(str(src), 10, cache=str(cache))
    assert e.nsamples == d.nsamples and torch.equal(e.feat_id[: e.nsamples], d.feat_id[: d.nsamples])
    # another nfields, a truncated file, foreign bytes: all ignored
    assert LibsvmDataset(str(src), 9, cache=str(cache)).feat_id.shape[1] == 9
    LibsvmDataset(str(src), 10, cache=str(cache))
    blob = cache.read_bytes()
    cache.write_bytes(blob[: len(blob) // 2])
    assert LibsvmDataset(str(src), 10, cache=str(cache)).nsamples == d.nsamples
    cache.write_bytes(b"not a cache at all")
    assert LibsvmDataset(str(src), 10, cache=str(cache)).nsamples == d.nsamples
