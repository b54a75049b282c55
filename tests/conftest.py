import os
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)
GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (gfx950); run with -m gpu on the GPU box")


def load_golden(name):
    with np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False) as z:
        return {k: z[k] for k in z.files}


@pytest.fixture(scope="session")
def golden():
    return load_golden


_SD_CACHE = {}


def state_dict_for(seed, style):
    """Seeded oracle weights (cached per session: 34 M parameters take ~1 s to draw)."""
    from oracle import weights
    key = (int(seed), str(style))
    if key not in _SD_CACHE:
        _SD_CACHE[key] = weights.make_state_dict(int(seed), str(style))
    return _SD_CACHE[key]


@pytest.fixture(scope="session")
def make_sd():
    return state_dict_for


# ---- two HIP runtimes share the GPU-test process: the library links /opt/rocm's libamdhip64, torch carries its own copy.
# Each runtime owns its queues / signals per process; a runtime that initialises AFTER the other one has created dozens of
# streams can find the device's per-process queue budget spent ("No HIP GPUs are available" from torch's lazy init after
# ~100 tests).  So (1) torch's runtime is initialised first, as in bench.py / sharded.py, and (2) every engine a test
# module leaves behind (module-level caches) is destroyed when the module finishes.
_LIVE_ENGINES = []


def pytest_sessionstart(session):
    try:
        import torch
        if torch.cuda.is_available():
            torch.cuda.init()
            torch.zeros(1, device="cuda").cpu()
    except Exception:
        pass
    try:
        from interactive_deep_colorization_amd import engine as _engine
        if not getattr(_engine.HipColorizer, "_tracked", False):
            _orig_init = _engine.HipColorizer.__init__

            def _init(self, *a, **kw):
                _orig_init(self, *a, **kw)
                _LIVE_ENGINES.append(self)
            _engine.HipColorizer.__init__ = _init
            _engine.HipColorizer._tracked = True
    except Exception:
        pass


@pytest.fixture(scope="module", autouse=True)
def _close_engines_left_by_the_module():
    yield
    while _LIVE_ENGINES:
        e = _LIVE_ENGINES.pop()
        try:
            e.close()
        except Exception:
            pass
    for modname in ("test_net_gpu",):
        import sys
        m = sys.modules.get(modname) or sys.modules.get("tests." + modname)
        if m is not None and hasattr(m, "_ENGINES"):
            m._ENGINES.clear()


def has_ab_partners():
    """True when the library was built with -DIDC_AB_PARTNERS (make EXTRA=-DIDC_AB_PARTNERS): the 32x32x16-MFMA partners conv_igemm_v2 / conv_ds_fused /
    conv1_1_bf16_kernel exist and `mfma16` / `ds_mfma16` = 0 select them.  The default library refuses those values (IDC_ERR_UNSUPPORTED): tests that
    compare a kernel with its partner run the partner leg only in that build (round 6)."""
    from interactive_deep_colorization_amd import _native, engine
    try:
        engine.set_option("mfma16", 0)
    except _native.IdcError:
        return False
    engine.set_option("mfma16", 1)
    return True
