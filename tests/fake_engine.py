"""Test double for the `rsx` engine module, backed by the CPU oracle.

Lets the HOST logic (src/indicies/*, src/search.py, sharded.py) be exercised on a machine without a
GPU.  It lives under tests/ and is never importable from the product.
"""
import pickle

import numpy as np

from oracle import oracle as o

METRIC_INNER_PRODUCT, METRIC_L2 = 0, 1


class _Base:
    def __init__(self, d):
        self.d = d
        self.x = np.zeros((0, d), dtype=np.float32)
        self.nprobe = 1
        self.metric_type = METRIC_INNER_PRODUCT

    @property
    def ntotal(self):
        return len(self.x)

    def add(self, x):
        self.x = np.concatenate([self.x, np.asarray(x, dtype=np.float32)], 0)


class IndexFlatIP(_Base):
    is_trained = True

    def search(self, q, k):
        return o.flat_search(np.asarray(q, np.float32), self.x, k, 0)


class IndexIVFFlat(_Base):
    def __init__(self, quantizer, d, nlist, metric=1):
        super().__init__(d)
        self.nlist, self.centroids = nlist, None

    @property
    def is_trained(self):
        return self.centroids is not None

    def train(self, x):
        x = np.asarray(x, np.float32)
        self.centroids = o.kmeans(0, x, self.nlist, 10, 1234)

    def _lm(self, payload):
        a, _ = o.assign_ip(self.centroids, self.x)
        return o.ListMajor(a, np.arange(len(self.x)), payload, self.nlist)

    def search(self, q, k):
        return o.ivfflat_search(0, self.centroids, self._lm(self.x), np.asarray(q, np.float32), self.nprobe, k)


class IndexIVFPQ(IndexIVFFlat):
    def __init__(self, quantizer, d, nlist, M, nbits, metric=1):
        super().__init__(quantizer, d, nlist, metric)
        self.M, self.codebooks = M, None

    @property
    def is_trained(self):
        return self.centroids is not None and self.codebooks is not None

    def train(self, x):
        super().train(x)
        x = np.asarray(x, np.float32)
        a, _ = o.assign_ip(self.centroids, x)
        self.codebooks = o.pq_train(o.residuals(self.centroids, x, a), self.M, 25, 1234)

    def search(self, q, k):
        a, _ = o.assign_ip(self.centroids, self.x)
        codes = o.pq_encode(self.codebooks, o.residuals(self.centroids, self.x, a))
        lm = o.ListMajor(a, np.arange(len(self.x)), codes, self.nlist)
        return o.ivfpq_search(self.centroids, self.codebooks, lm, np.asarray(q, np.float32), self.nprobe, k)


def write_index(index, path):
    with open(path, "wb") as f:
        pickle.dump(index, f)


def read_index(path, device=None):
    with open(path, "rb") as f:
        return pickle.load(f)
