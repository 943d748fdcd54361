"""One rank of tests/test_multirank_gpu.py: the real CLI (real scans on the GPU) with a gloo
communicator injected in place of RCCL, which refuses two ranks on the one GPU of the test
box.  argv: output file for rank 0's stdout, then the CLI's own arguments."""
import io
import os
import sys
from contextlib import redirect_stdout

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["PFA_ALLOW_SHARED_GPU"] = "1"       # both ranks bind GPU 0

from polyfasta_amd import cli  # noqa: E402
from tests._gloo_comm import GlooComm  # noqa: E402

buf = io.StringIO()
try:
    with redirect_stdout(buf):
        rc = cli.main(sys.argv[2:], comm=GlooComm())
finally:      # (also when the job dies where the reference dies: what rank 0 printed until then)
    if os.environ.get("RANK", "0") == "0":
        open(sys.argv[1], "w").write(buf.getvalue())
    else:
        assert buf.getvalue() == "", "only rank 0 prints"
sys.exit(rc)
