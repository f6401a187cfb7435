"""TEST INFRASTRUCTURE -- the Markov-cluster loop of the reference through scipy on the CPU, the oracle for the device loop
(swiftortho_amd/csrc/mcl.hip, so_mcl).  Follows /root/reference/bin/find_cluster.py `normalize` (636-646) and `mcl` (652-689)
statement by statement on purpose; never imported by the product (tests/test_abi.py checks).  Pinned by the `clu_*` goldens
(stdout of the REAL find_cluster.py): tests/test_find_cluster.py runs the product's host bookkeeping with this loop plugged in."""
import numpy as np
from scipy import sparse


def _normalize(x):
    y = np.asarray(x.sum(0))[0]
    if y.min() == 0 and y.max() > 0:
        y += y.nonzero()[0].min() / 1e3
    else:
        y += 1e-8
    x.data /= y.take(x.indices, mode='clip')


def scipy_mcl(indptr, indices, data, inflation, expansion=2, prune=1e-5, rtol=1e-5, atol=1e-8, rounds=100, check=5):
    """same signature and result as swiftortho_amd.find_cluster.device_mcl: CSR in, final CSR (storage order, stored zeros) out"""
    n = len(indptr) - 1
    x = sparse.csr_matrix((np.asarray(data, dtype=np.float32), np.asarray(indices, dtype=np.int32), np.asarray(indptr)), shape=(n, n), dtype='float32')
    for i in range(rounds):
        _normalize(x)
        if i % check == 0:
            x_old = x.copy()
        x = x ** expansion
        x.data **= inflation
        if i % check == 0 and i > 0:
            if (abs(x - x_old) - rtol * abs(x_old)).max() <= atol:
                break
        x.data[x.data < prune] = 0.
    return np.asarray(x.indptr, dtype=np.int64), np.asarray(x.indices, dtype=np.int32), np.asarray(x.data, dtype=np.float32)
