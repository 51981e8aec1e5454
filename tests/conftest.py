import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    if os.environ.get("DCTR_POISON_EMPTY"):
        # debugging aid: every torch.empty() comes back filled with NaN / the largest integer, so that a kernel reading a buffer nobody
        # wrote shows up as a wrong result / an index-out-of-range flag instead of depending on what the allocator hands out
        import torch
        torch.use_deterministic_algorithms(True, warn_only=True)
        torch.utils.deterministic.fill_uninitialized_memory = True


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session")
def device():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no HIP device")
    return torch.device("cuda:0")


def pytest_terminal_summary(terminalreporter):
    """how many compared elements needed more than north_star's literal bar (1e-4 of the result), i.e. leaned on
    the few-ulp-of-the-summed-magnitude term of tests/util.assert_close_terms."""
    from tests.util import TERMS_REPORT
    if not TERMS_REPORT:
        return
    n = sum(r[1] for r in TERMS_REPORT)
    out = sum(r[2] for r in TERMS_REPORT)
    tr = terminalreporter
    tr.write_sep("-", "assert_close_terms: %d comparisons, %d elements, %d (%.4f %%) outside 1e-4 * |ref| alone" % (
        len(TERMS_REPORT), n, out, 100.0 * out / max(n, 1)))
    for what, size, o, worst_lit, worst in sorted(TERMS_REPORT, key=lambda r: -r[2])[:12]:
        if o:
            tr.write_line("  %-48s %8d elements, %6d outside (%.3f %%), worst err = %.2f x (1e-4 |ref|) = %.2f x bar" % (
                what[:48], size, o, 100.0 * o / max(size, 1), worst_lit, worst))
