"""The oracle is test infrastructure: nothing under fetalsyngen_amd/ may import or execute it, and the
product has no torch/numpy compute fallback for the kernels."""
import re
from pathlib import Path

REPO = Path(__file__).resolve().parent.parent


def test_product_never_references_oracle():
    for p in (REPO / "fetalsyngen_amd").rglob("*.py"):
        text = p.read_text()
        assert not re.search(r"^\s*(from|import)\s+oracle\b", text, flags=re.M), p
        assert "fsg_oracle" not in text, p


def test_bench_uses_oracle_only_for_cpu_baseline():
    text = (REPO / "bench.py").read_text()
    uses = [m.start() for m in re.finditer(r"fsg_oracle", text)]
    assert uses, "bench.py must time the oracle as cpu_baseline"
    body = text[text.index("def cpu_baseline"):]
    end = re.search(r"^def ", body[4:], flags=re.M)
    span = (text.index("def cpu_baseline"), text.index("def cpu_baseline") + 4 + (end.start() if end else len(body)))
    for u in uses:
        assert span[0] <= u < span[1], "oracle referenced outside cpu_baseline()"


def test_kernel_wrappers_have_no_torch_math():
    """kernels.py is pointer plumbing: no torch arithmetic that could act as a silent fallback."""
    text = (REPO / "fetalsyngen_amd" / "kernels.py").read_text()
    for banned in ("torch.nn.functional", "conv3d", "grid_sample", "torch.exp(", "torch.pow("):
        assert banned not in text, banned
