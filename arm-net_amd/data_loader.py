"""data_loader — native input pipeline for the libsvm datasets (drop-in for the reference's
data_loader.py:12-73).

``LibsvmDataset(fname, nfields)`` keeps the reference's attributes (feat_id Long[N,F], feat_value Float[N,F],
y Float[N], nsamples) and its skip-malformed-lines behaviour, but the text is parsed by
csrc/libsvm_reader.cpp (mmap + OpenMP) instead of a Python loop over lines.
``libsvm_dataloader(args)`` returns the same three torch DataLoaders the reference's train.py expects.
``LibsvmDataset(..., cache=True | path)`` keeps the parsed split beside the text as ONE binary file (header + the three
arrays, memory-mapped back: SURVEY.md §8f-3's "binary pre-tokenised format"); a cache whose
header does not match the text file's size / mtime / nfields is ignored and rewritten.
``DeviceLoader`` is the MI355X-first alternative: the whole split lives in HBM (Criteo's 45 M x 39 samples
are 21 GB of the 288 GB) and batches are slices — no worker processes, no pinned staging, no per-batch H2D.
"""
import ctypes
import glob
import os

import numpy as np
import torch
from torch.utils.data import DataLoader, Dataset

_LIB = os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib", "libarmnet_io.so")
_io = None


def _lib():
    global _io
    if _io is None:
        if not os.path.exists(_LIB):
            raise RuntimeError(f"{_LIB} is missing: run `python -c 'import __graft_entry__ as g; g.build()'`")
        _io = ctypes.CDLL(_LIB)
        _io.armnet_libsvm_count_lines.restype = ctypes.c_int64
        _io.armnet_libsvm_parse.restype = ctypes.c_int64
    return _io


_CACHE_MAGIC = b"ARMNETDS"
_CACHE_VERSION = 1


def _cache_header(st, nfields, nlines, nsamples, nskipped):
    """st: os.stat of the text file taken BEFORE it was parsed (a file rewritten during the parse must not get a header
    that validates what was parsed from its previous content: round-4 advisor finding)"""
    return np.array([_CACHE_VERSION, nfields, nlines, nsamples, nskipped, st.st_size, st.st_mtime_ns], dtype=np.int64)


def _cache_read(cache, fname, nfields):
    """(feat_id, feat_value, y, nsamples, nskipped) from a cache written for exactly this text file, or None"""
    try:
        with open(cache, "rb") as f:
            if f.read(8) != _CACHE_MAGIC:
                return None
            h = np.fromfile(f, dtype=np.int64, count=7)
            st = os.stat(fname)
            if h.size != 7 or h[0] != _CACHE_VERSION or h[1] != nfields or h[5] != st.st_size or h[6] != st.st_mtime_ns:
                return None
            rows = max(int(h[2]), 1)
            off = 8 + 7 * 8
            need = off + rows * nfields * 12 + rows * 4
            if os.fstat(f.fileno()).st_size != need:
                return None                                     # truncated / foreign file
        # copy-on-write maps: construction is O(1), pages come in as they are touched (a DeviceLoader streams them once),
        # and the tensors stay writable like the parsed ones without ever changing the file
        ids = np.memmap(cache, dtype=np.int64, mode="c", offset=off, shape=(rows, nfields))
        vals = np.memmap(cache, dtype=np.float32, mode="c", offset=off + rows * nfields * 8, shape=(rows, nfields))
        y = np.memmap(cache, dtype=np.float32, mode="c", offset=off + rows * nfields * 12, shape=(rows,))
    except (OSError, ValueError):
        return None
    return torch.from_numpy(ids), torch.from_numpy(vals), torch.from_numpy(y), int(h[3]), int(h[4])


def _cache_write(cache, header, feat_id, feat_value, y):
    tmp = f"{cache}.tmp{os.getpid()}"
    try:
        with open(tmp, "wb") as f:
            f.write(_CACHE_MAGIC)
            header.tofile(f)
            feat_id.numpy().tofile(f)
            feat_value.numpy().tofile(f)
            y.numpy().tofile(f)
        os.replace(tmp, cache)                                  # readers never see a half-written cache
    except OSError as e:
        print(f"# binary cache {cache} not written: {e}")
        try:
            os.remove(tmp)
        except OSError:
            pass


class LibsvmDataset(Dataset):
    """Dataset loader for the libsvm text format (reference: data_loader.py:12-55).  cache: None / False = parse the text
    every time (the reference's behaviour); True = `<fname>.armnet.bin`; a path = that file."""

    def __init__(self, fname, nfields, nthreads=0, cache=None):
        cache_path = (f"{fname}.armnet.bin" if cache is True else cache) or None
        if cache_path and os.path.exists(cache_path):
            hit = _cache_read(cache_path, fname, int(nfields))
            if hit is not None:
                self.feat_id, self.feat_value, self.y, self.nsamples, self.nskipped = hit
                print(f"# {self.nsamples} data samples loaded... (binary cache {cache_path})")
                return
        lib = _lib()
        path = os.fsencode(fname)
        st_before = os.stat(fname) if cache_path else None
        nlines = int(lib.armnet_libsvm_count_lines(path))
        if nlines < 0:
            raise FileNotFoundError(fname)
        feat_id = torch.empty(max(nlines, 1), nfields, dtype=torch.int64)
        feat_value = torch.empty(max(nlines, 1), nfields, dtype=torch.float32)
        y = torch.empty(max(nlines, 1), dtype=torch.float32)
        n_bad = ctypes.c_int64(0)
        n = int(lib.armnet_libsvm_parse(path, int(nfields), ctypes.c_int64(nlines),
                                        ctypes.c_void_p(feat_id.data_ptr()), ctypes.c_void_p(feat_value.data_ptr()),
                                        ctypes.c_void_p(y.data_ptr()), ctypes.byref(n_bad), int(nthreads)))
        if n < 0:
            raise OSError(f"cannot parse {fname} (code {n})")
        self.nsamples = n
        self.nskipped = int(n_bad.value)
        # the reference preallocates one row per text line and fills the first nsamples
        self.feat_id, self.feat_value, self.y = feat_id, feat_value, y
        if self.nskipped:
            print(f"{self.nskipped} line(s) of incorrect data format skipped !")
        print(f"# {self.nsamples} data samples loaded...")
        if cache_path:
            # rows at and past nsamples (skipped lines) are uninitialised host memory: zero them, so that the cache bytes
            # are a function of the text file alone and nothing of this process leaks into a file
            feat_id[n:].zero_()
            feat_value[n:].zero_()
            y[n:].zero_()
            st_after = os.stat(fname)
            if (st_after.st_size, st_after.st_mtime_ns) == (st_before.st_size, st_before.st_mtime_ns):
                _cache_write(cache_path, _cache_header(st_before, int(nfields), nlines, self.nsamples, self.nskipped),
                             feat_id, feat_value, y)
            else:
                print(f"# {fname} changed while it was parsed: binary cache not written")

    def __len__(self):
        return self.nsamples

    def __getitem__(self, idx):
        return {"id": self.feat_id[idx], "value": self.feat_value[idx], "y": self.y[idx]}


def libsvm_dataloader(args):
    """reference: data_loader.py:57-73 — train/valid/test DataLoaders from <data_dir><dataset>/{tr,va,te}*libsvm."""
    data_dir = args.data_dir + args.dataset
    files = [glob.glob(f"{data_dir}/{p}*libsvm")[0] for p in ("tr", "va", "te")]
    shuffle = (True, False, False)
    cache = getattr(args, "binary_cache", None)       # not a reference flag: True keeps <file>.armnet.bin beside each split
    return tuple(DataLoader(LibsvmDataset(f, args.nfield, cache=cache), batch_size=args.batch_size, shuffle=s,
                            num_workers=args.workers, pin_memory=True) for f, s in zip(files, shuffle))


class DeviceLoader:
    """Batches of a LibsvmDataset that lives entirely in device memory.  Iterating yields the same
    {'id','value','y'} dicts as the reference's DataLoader, already on the GPU."""

    def __init__(self, dataset, batch_size, shuffle=False, device="cuda", drop_last=False, seed=0):
        n = dataset.nsamples
        self.ids = dataset.feat_id[:n].to(device)
        self.vals = dataset.feat_value[:n].to(device)
        self.y = dataset.y[:n].to(device)
        self.n, self.batch_size, self.shuffle, self.drop_last = n, int(batch_size), shuffle, drop_last
        self.gen = torch.Generator(device=self.ids.device).manual_seed(seed)

    def __len__(self):
        return self.n // self.batch_size if self.drop_last else (self.n + self.batch_size - 1) // self.batch_size

    def __iter__(self):
        order = torch.randperm(self.n, device=self.ids.device, generator=self.gen) if self.shuffle else None
        for i in range(len(self)):
            lo, hi = i * self.batch_size, min((i + 1) * self.batch_size, self.n)
            if order is None:
                # value is clamped in place by the model: hand out a copy so the dataset stays pristine
                yield {"id": self.ids[lo:hi], "value": self.vals[lo:hi].clone(), "y": self.y[lo:hi]}
            else:
                sel = order[lo:hi]
                yield {"id": self.ids[sel], "value": self.vals[sel], "y": self.y[sel]}
