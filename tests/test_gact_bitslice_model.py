"""The CPU model of the bit-sliced GACT kernel (tests/models/gact_bitslice_model.c) against the oracle.

The model runs, for one lane, exactly the operations gact_bs_kernel runs: difference planes, streamed
sequence windows with sentinels, checkpoints every BS_K anti-diagonals, per-block recompute and walk.
Keeping it green on the CPU pins the kernel's formulation without a GPU; the GPU parity tests
(test_gpu_parity.py, LRM_GACT_IMPL=4) then only have to show that the kernel matches its model's answers."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

import orc

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "models", "gact_bitslice_model.c")
LIB = os.path.join(HERE, "models", "libgact_bitslice_model.so")


@pytest.fixture(scope="module")
def model():
    if not os.path.exists(LIB) or os.path.getmtime(LIB) < os.path.getmtime(SRC):
        subprocess.check_call(["gcc", "-O2", "-shared", "-fPIC", "-o", LIB, SRC])
    lib = C.CDLL(LIB)
    lib.bsm_gact.restype = C.c_int
    lib.bsm_gact.argtypes = [C.c_char_p, C.c_int, C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p,
                             C.POINTER(C.c_int)]

    def run(q, d, T, O, extra=0, W=128):
        ops = np.zeros(len(q) + len(d) + 8, dtype=np.uint8)
        n_ops = C.c_int()
        score = lib.bsm_gact(q, len(q), d, len(d), T, O, W, extra, ops.ctypes.data, C.byref(n_ops))
        return score, bytes(ops[:n_ops.value])
    return run


def _noisy(rng, d, n, err):
    q = bytearray()
    src = int(rng.integers(0, 20))
    while len(q) < n:
        r = rng.random()
        if r < err / 3:
            q.append(rng.choice(list(b"ACGT")))
        elif r < 2 * err / 3:
            src += 1
        else:
            q.append(rng.choice(list(b"ACGT")) if (r < err or src >= len(d)) else d[src])
            src += 1
    return bytes(q)


@pytest.mark.parametrize("W", [128, 64, 32, 66, 20, 2])
@pytest.mark.parametrize("T,O", [(320, 120), (512, 120), (512, 0), (100, 99), (64, 16), (33, 7), (16, 0), (200, 40)])
def test_model_equals_oracle(model, T, O, W):
    rng = np.random.default_rng(T * 7 + O + W)
    for it in range(60 if W == 128 else 25):
        m = int(rng.integers(1, 2500 if it % 6 == 0 else 700))
        d = bytes(rng.choice(list(b"ACGT"), size=m).astype(np.uint8))
        n = m if it % 3 == 0 else int(rng.integers(1, 900))
        q = _noisy(rng, d, n, float(rng.integers(0, 30)) / 100)
        extra = int(rng.integers(0, 3)) * 32              # a wave-wide start above this lane's own tile corner
        want = orc.gact(q, d, T, O, W)
        assert model(q, d, T, O, extra, W) == (want[0], want[1]), (it, n, m)


def test_model_rejects_other_bytes(model):
    assert model(b"ACGTN", b"ACGTA", 320, 120)[0] == -1      # the kernel routes such reads to the byte kernels
