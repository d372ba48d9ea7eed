"""TEST-ONLY rank program for the `rnascan --gpus N` launcher test on a machine without a GPU: the CLI with the
oracle-backed test engine (tests/engines.py) in place of the HIP engine.  The product's rank program is
`python -m rnascan_amd`; tests point rnascan_amd.cli.RANK_COMMAND here."""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))

from engines import OracleEngine  # noqa: E402
from rnascan_amd import cli  # noqa: E402

if os.environ.get("RNASCAN_TEST_FAIL_RANK") == os.environ.get("RANK"):
    sys.exit(5)
rc = cli.main(sys.argv[1:], engine=OracleEngine(), out=sys.stdout)
sys.stdout.flush()
sys.exit(rc)
