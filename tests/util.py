"""Shared helpers for the parity tests."""
import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

# north_star tolerance: outputs within 1e-3 relative, fp32.
REL_TOL = 1e-3
# gradient gate of the kernel-level / small-fixture tests: ~4x the worst case measured over every comparison of test_gpu_tsf.py,
# test_gpu_effnet.py and test_gpu_xception.py on an MI355X (4.4e-5, round 6); the full-size gates live in test_gpu_e2e.py
GRAD_TOL_UNIT = 2e-4


def golden(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"))


def rel_err(got, ref):
    """max|got-ref| / max|ref|  (the per-tensor gate of SURVEY.md §8c)."""
    got = torch.as_tensor(np.asarray(got)).double() if not torch.is_tensor(got) else got.detach().cpu().double()
    ref = torch.as_tensor(np.asarray(ref)).double() if not torch.is_tensor(ref) else ref.detach().cpu().double()
    assert got.shape == ref.shape, (got.shape, ref.shape)
    den = ref.abs().max().item()
    return (got - ref).abs().max().item() / max(den, 1e-30)


def assert_close(got, ref, tol=REL_TOL, what=""):
    e = rel_err(got, ref)
    if os.environ.get("MT_TEST_VERBOSE"):
        print(f"[assert_close] {what}: rel err {e:.3e} (tol {tol:.1e})")
    assert e <= tol, f"{what}: rel err {e:.3e} > {tol:.1e}"
    return e


def checksum(t):
    return float(t.double().sum())


def dropout_multipliers(g, batch, tokens, dim):
    """{(layer, kind): keep / (1 - p) multiplier tensor} from the bit-packed keep masks of tests/golden/tsf_dropout.npz
    (kind 0 / 1: after the time / space attention's output projection, [B, N, dim]; kind 2: between GEGLU and net.3, [B, N, 4 dim])."""
    out = {}
    for li in range(int(g["depth"])):
        for kind in range(3):
            width = 4 * dim if kind == 2 else dim
            p = float(g["ff_p"] if kind == 2 else g["attn_p"])
            bits = np.unpackbits(g[f"keep.{li}.{kind}"])[:batch * tokens * width]
            out[(li, kind)] = torch.from_numpy(bits.astype(np.float32)).reshape(batch, tokens, width) / (1.0 - p)
    return out


def probe_vector(name, n, seed=0):
    """Fixed pseudo-random weights r in [-1, 1) for the whole-tensor checksum pair (sum g, sum g*r) of a parameter gradient:
    numpy Philox keyed by (seed, crc32(name)) -- the same vector in tools/make_golden.py (reference side) and in the GPU tests."""
    import zlib
    rng = np.random.Generator(np.random.Philox(key=[int(seed), zlib.crc32(name.encode())]))
    return rng.uniform(-1.0, 1.0, int(n))
