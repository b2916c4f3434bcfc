import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def oracle_mod():
    import oracle
    oracle.build()
    return oracle


@pytest.fixture(scope="session")
def vrt_ctx():
    """One context on GPU 0 for the whole GPU test session."""
    # torch brings its own copy of the HIP runtime; tests that hand torch tensors to libvrt need
    # torch's to come up first (the other order leaves torch without a device), as in bench.py
    import torch
    if torch.cuda.is_available():
        torch.cuda.init()
    from raytracing_amd import host
    ctx = host.Context(0)
    yield ctx
    ctx.close()


OPTION_DEFAULTS = dict(xcd_remap=0, sample_shift=-1, wf_defer=1, wf_refill=1, wf_max_slots=0, wf_any_cap=-1, wf_queue_cap=0,
                       builder=-1, collapse_parity=0, no_big=0, packet=-1, beam=-1, block_compact=0, quad_max=-1, stack_cap=0, slab_fma=1, fail_after_build=0)


@pytest.fixture
def vrt_opts(vrt_ctx):
    """set(name, value) changes a study option of the shared context; every option is back at its
    default when the test ends."""
    def set_option(name, value):
        vrt_ctx.set_option(name, value)
    yield set_option
    for k, v in OPTION_DEFAULTS.items():
        vrt_ctx.set_option(k, v)


def assert_frame_matches_oracle(rgba, hit, rays, oracle_frame, tol=1.0 / 255.0):
    """The parity bar of every frame test: primary hit ids and per-class ray counts bit-exact, RGB
    within 1/255 of the oracle's frame (NaN where the oracle has NaN)."""
    import numpy as np
    o_rgba, o_hit, o_rays = oracle_frame
    if hit is not None:
        assert np.array_equal(hit, o_hit), f"{np.count_nonzero(hit != o_hit)} primary hit ids differ from the oracle"
    if rays is not None:
        assert np.array_equal(np.array(list(rays), np.uint64), o_rays), "ray counts differ from the oracle"
    both_nan = np.isnan(rgba) & np.isnan(o_rgba)
    err = np.abs(np.where(both_nan, 0, rgba) - np.where(both_nan, 0, o_rgba))
    assert not np.isnan(err).any() and err.max() <= tol, float(err.max())
