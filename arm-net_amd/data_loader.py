"""data_loader — native input pipeline for the libsvm datasets (drop-in for the reference's
data_loader.py:12-73).

``LibsvmDataset(fname, nfields)`` keeps the reference's attributes (feat_id Long[N,F], feat_value Float[N,F],
y Float[N], nsamples) and its skip-malformed-lines behaviour, but the text is parsed by
csrc/libsvm_reader.cpp (mmap + OpenMP) instead of a Python loop over lines.
``libsvm_dataloader(args)`` returns the same three torch DataLoaders the reference's train.py expects.
``DeviceLoader`` is the MI355X-first alternative: the whole split lives in HBM (Criteo's 45 M x 39 samples
are 21 GB of the 288 GB) and batches are slices — no worker processes, no pinned staging, no per-batch H2D.
"""
import ctypes
import glob
import os

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


class LibsvmDataset(Dataset):
    """Dataset loader for the libsvm text format (reference: data_loader.py:12-55)."""

    def __init__(self, fname, nfields, nthreads=0):
        lib = _lib()
        path = os.fsencode(fname)
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

    def __len__(self):
        return self.nsamples

    def __getitem__(self, idx):
        return {"id": self.feat_id[idx], "value": self.feat_value[idx], "y": self.y[idx]}


def libsvm_dataloader(args):
    """reference: data_loader.py:57-73 — train/valid/test DataLoaders from <data_dir><dataset>/{tr,va,te}*libsvm."""
    data_dir = args.data_dir + args.dataset
    files = [glob.glob(f"{data_dir}/{p}*libsvm")[0] for p in ("tr", "va", "te")]
    shuffle = (True, False, False)
    return tuple(DataLoader(LibsvmDataset(f, args.nfield), batch_size=args.batch_size, shuffle=s,
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
