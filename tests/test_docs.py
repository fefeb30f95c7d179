"""Docs hygiene on the CPU: the documents that describe the current state cite evidence that exists and is of the current
round (tools/check_docs.py has the rules and the short list of older measurements DESIGN.md may still quote, each with its reason)."""
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_documents_cite_existing_evidence_of_the_current_round():
    out = subprocess.run([sys.executable, os.path.join(REPO, "tools", "check_docs.py")], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout[-4000:]


def test_header_and_documents_agree_on_the_abi_version_and_option_keys():
    """INTEGRATION.md records the one ABI break of the library's life ("auto_geometry" removed in round 5, back as a deprecated alias in round 6,
    version string 0.2), and every option key of scv_set_option is in the header (tests/test_abi_symbols.py) AND spelled as in the source."""
    integ = open(os.path.join(REPO, "INTEGRATION.md")).read()
    src = open(os.path.join(REPO, "o1_inference_scaling_laws_amd", "csrc", "scvote.hip")).read()
    assert '"scvote 0.2 (gfx950)"' in src and "scvote 0.2" in integ and "auto_geometry" in integ
    assert "SCV_FLAG_PACKED_CELLS" in integ and "budgets_host" in integ
