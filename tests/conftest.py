import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")
    config.addinivalue_line("markers", "reference: needs /root/reference (build container only)")
    config.addinivalue_line("markers", "probes: covers a tools-build experiment (libfedcola_hip_probes.so): runs under FC_PROBES_LIB=1, which "
                                       "tests/test_gpu_model.py::test_tools_build_experiments does in a child process")
    # the C-ABI library is a build artefact (git-ignored): make sure it exists and is current before any test loads it
    try:
        from fedcola_amd import build as _b
        import torch
        # missing: always build.  stale (sources newer than the .so): rebuild only in the GPU-less build container -- a GPU box
        # receives a file snapshot whose timestamps say nothing, and the .so travels with it
        if os.path.exists(_b.HIPCC) and (not os.path.exists(_b.OUT) or (_b.needs_build() and not torch.cuda.is_available())):
            _b.build(force=False, verbose=False)
    except Exception as e:  # a failed build surfaces in the tests that need the library
        print(f"[conftest] building libfedcola_hip.so failed: {e}", file=sys.stderr)


def pytest_collection_modifyitems(config, items):
    import torch
    has_gpu = torch.cuda.is_available()
    skip_gpu = pytest.mark.skip(reason="no GPU")
    skip_probes = pytest.mark.skip(reason="tools-build experiment: run with FC_PROBES_LIB=1 (test_tools_build_experiments does)")
    for item in items:
        if "gpu" in item.keywords and not has_gpu:
            item.add_marker(skip_gpu)
        if "probes" in item.keywords and not os.environ.get("FC_PROBES_LIB"):
            item.add_marker(skip_probes)
