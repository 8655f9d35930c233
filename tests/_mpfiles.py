"""rank -> result hand-over of the multi-process tests through files.

A ``multiprocessing.Manager().dict()`` is served by a process FORKED from the pytest process, i.e. from a process that has
initialised the GPU, and receives tensors as shared-memory descriptors passed between processes.  On the GPU box that server was
found dead (EOFError in the sending rank) whenever the full-size step cases had run earlier in the same pytest process; the ranks
themselves had finished their work.  Files need no third process."""
import os
import shutil
import tempfile

import torch


class FileDict:
    def __init__(self):
        self.path = tempfile.mkdtemp(prefix="dsvgp_mp_")

    def _file(self, key):
        return os.path.join(self.path, "%s.pt" % (key,))

    def __setitem__(self, key, value):
        tmp = self._file(key) + ".tmp"
        torch.save(value, tmp)
        os.replace(tmp, self._file(key))

    def __getitem__(self, key):
        if not os.path.exists(self._file(key)):
            raise KeyError(key)
        return torch.load(self._file(key), weights_only=False)

    def __contains__(self, key):
        return os.path.exists(self._file(key))

    def collect(self, keys):
        """{key: value} for the given keys; the directory is removed afterwards"""
        out = {k: self[k] for k in keys}
        shutil.rmtree(self.path, ignore_errors=True)
        return out
